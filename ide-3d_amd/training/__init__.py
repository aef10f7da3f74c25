"""Render-path half of the reference `training` package (volumetric renderer, StyleGAN2 blocks, tri-plane generator)."""
