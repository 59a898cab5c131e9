"""MI355X-native audio-reactive StyleGAN2 inference path (drop-in for JCBrouwer/maua-stylegan2's
generate_audiovisual.py / generate() surface).  See DESIGN.md."""
__version__ = "0.1.0"
