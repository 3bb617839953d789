import sys, time, os
sys.path.insert(0, "."); sys.path.insert(0, "generative-turbulence_amd")
import torch, bench
dev = torch.device("cuda:0")
for amp in (True, False):
    t0 = time.time()
    r = bench.torch_rocm_baseline(6, dev, amp, steps=1)
    print(f"amp={amp} wall {time.time()-t0:.1f} s  ms_per_step {r['ms_per_step']:.1f}  MIOPEN_FIND_MODE={os.environ.get('MIOPEN_FIND_MODE')}", flush=True)
