import os, sys
sys.path.insert(0, "tools"); sys.path.insert(0, "."); sys.path.insert(0, "generative-turbulence_amd")
import soak_train as S
res = {}
for sw in ("0", "1", "1"):
    os.environ["TDX_WGRAD_STREAM"] = sw
    l, dt, mem = S.run("bf16", 400)
    res.setdefault(sw, []).append(l)
    print(f"TDX_WGRAD_STREAM={sw}: {dt:.2f} ms/step, peak {mem:.1f} GB, loss[0]={l[0]:.6f} loss[199]={l[199]:.6f} loss[399]={l[399]:.6f}", flush=True)
a, b, c = res["0"][0], res["1"][0], res["1"][1]
dev = lambda u, v: max(abs(x - y) / abs(x) for x, y in zip(u, v))
print(f"largest relative loss deviation over 400 steps: side stream vs launching stream {dev(a, b):.2e}; side stream run-to-run {dev(b, c):.2e}")
