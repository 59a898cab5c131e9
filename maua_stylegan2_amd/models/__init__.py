"""Mirror of the reference's ``models`` package: only the StyleGAN2 generator side is on the hot path."""
