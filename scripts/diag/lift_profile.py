"""where lift_flat spends its time on the 10 M-edge cfg-5 graph (device-assisted path)"""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'lifted-hybrid-variational-inference_amd')]
import numpy as np, torch
from lhvi import _abi, synth, lifting
flat, sym, rv0, f0 = synth.rgm_structured_flat()
dg = _abi.DeviceGraph(flat)
rvc, fc = lifting.refine_flat(flat, sym, rv0, f0, dg=dg, device_out=True)
lifting.lift_flat(flat, rvc, fc, dg=dg)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    lifting.lift_flat(flat, rvc, fc, dg=dg)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
