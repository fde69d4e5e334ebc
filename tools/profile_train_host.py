"""Host-side profile of the training step (cProfile, cumulative): where the python / sync time of head.loss goes."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], '2', '64', '1']
import runpy
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'time_train.py'))
step = ns['step']
import torch
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
print(s.getvalue()[:9000])
