"""Every environment switch of the Python layer, read ONCE when the package is imported (never inside a forward pass).

Defaults are the product; each switch exists for A/B timing of one design decision and has a module attribute that overrides it per
object.  The library's own switches (C side) are `ORP_DCN_SPLIT` (arithmetic of the fp32 contractions: 3 default, 6, 9, 0 = exact fp32
MFMA; `orp_dcn_set_split_mode`), `ORP_DCNS_MT` (tile height of the split kernel, dev aid), `ORP_DCN_KSPLIT` (tap-granular split of
the exact-fp32 DeformConv launch) and `ORP_FILL=memset` (fills as hipMemsetAsync instead of kernels: the A/B aid of DESIGN.md 4.5).
The table in DESIGN.md section 0 lists them all with what they default to and why.
"""
import os


def _on(name, default='1'):
    return os.environ.get(name, default) == '1'


TOWER_SPLIT = _on('ORP_TOWER_SPLIT')          # head towers on the split matrix-pipe kernel (head.split_towers overrides)
FPN_SPLIT = _on('ORP_FPN_SPLIT')              # FPN output convolutions on it (neck.split_convs overrides)
TOWER_GN_FUSE = _on('ORP_TOWER_GN_FUSE')      # GroupNorm fused around the tower convolutions (head.fuse_tower_norm overrides)
TRAIN_SPLIT = _on('ORP_TRAIN_SPLIT')          # training: tower / FPN convolutions as conv_split_train nodes
DETERMINISTIC = _on('ORP_DETERMINISTIC', '0')  # training: fixed-order DeformConv backward for both branches (head.deterministic_backward overrides)
