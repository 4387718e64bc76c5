"""SPOT tracker hot path on MI355X: host-side mirror of the reference's model / actor / trainer / tracker API over the
C-ABI library libspot_hip.so (hand-written HIP kernels for gfx950).  GPU only: spot_amd.lib raises if the library is
missing; nothing here falls back to a CPU path."""
