"""Host-side mirror of the reference's `torch_utils` package for the IDE-3D render path (MI355X build)."""
