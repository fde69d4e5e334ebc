"""Host-side mirror of the DOTA_devkit modules on the hot path (DOTA_devkit/poly_nms_gpu)."""
