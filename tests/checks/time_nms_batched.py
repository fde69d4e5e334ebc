"""Development aid (GPU box): the (image x class)-batched rotated NMS of bench.py's `nms_batched_16_images` line."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
print(json.dumps(bench.batched_nms_line(torch.device('cuda:0')), indent=1))
