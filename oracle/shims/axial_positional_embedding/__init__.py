"""Stand-in for the un-vendored `axial_positional_embedding` package, so that the UNMODIFIED reference can run with `add_pos_emb=True` in the
build container.  TEST INFRASTRUCTURE.  The class is the repo's own restatement (transfusion_pytorch_amd/axial.py - parity UNPINNED for the
MLP's arithmetic, see there): with both sides using it, the goldens pin everything AROUND the embedding (where it is added, the projected
axial shapes it is evaluated on, its gradient) to the reference's code."""
from transfusion_pytorch_amd.axial import ContinuousAxialPositionalEmbedding  # noqa: F401
