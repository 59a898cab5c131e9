"""Oracle (test infrastructure, CPU) for the default audio-reactive plugin: restates
/root/reference/audioreactive/examples/default.py:12-45 on top of signal_oracle (gaussian_filter :319-368,
chroma_weight_latents latent.py:15-26).  Pinned, through tests/golden/default_plugin.npz, to what the reference's own
callbacks produce (tests/test_oracle_golden.py::test_plugin_oracle_matches_reference_plugin).  The two random fields of
get_noise are drawn by the caller-supplied ``randn`` (the reference uses th.randn on the device)."""
from . import signal_oracle as so


def get_latents(selection, chroma, lo_onsets, hi_onsets, smf=1.0):
    """default.py:12-26: chroma-weighted latents, sigma 4 smoothing, onset cross-fades toward selection[-4] / [-7], causal sigma 2."""
    latents = so.gaussian_filter(so.chroma_weight_latents(chroma, selection), 4, smf=smf)
    lo, hi = lo_onsets[:, None, None], hi_onsets[:, None, None]
    latents = hi * selection[[-4]] + (1 - hi) * latents
    latents = lo * selection[[-7]] + (1 - lo) * latents
    return so.gaussian_filter(latents, 2, causal=0.2, smf=smf)


def get_noise(height, width, n_frames, lo_onsets, hi_onsets, randn, smf=1.0):
    """default.py:28-45: None above 256 px; sigma-5 and sigma-128 filtered Gaussian fields, bass blend below 128 px, treble blend
    above 32 px, normalised to std 1 / 2.5."""
    if width > 256:
        return None
    lo, hi = lo_onsets[:, None, None, None], hi_onsets[:, None, None, None]
    noise_noisy = so.gaussian_filter(randn((n_frames, 1, height, width)), 5, smf=smf)
    noise = so.gaussian_filter(randn((n_frames, 1, height, width)), 128, smf=smf)
    if width < 128:
        noise = lo * noise_noisy + (1 - lo) * noise
    if width > 32:
        noise = hi * noise_noisy + (1 - hi) * noise
    return noise / (noise.std() * 2.5)
