import sys, os, json, torch
sys.path.insert(0, os.getcwd())
import recnext_amd
from recnext_amd import ops, _lib
from tools.bench_blocks import time_fresh, time_fn
torch.manual_seed(0)
for n in (2, 8, 32, 64, 128):
    for c, h, lv in ((128, 64, 3), (64, 128, 4)):
        if n * c * h * h > 128 * 128 * 64 * 64: continue
        mod = recnext_amd.RecConv2d(c, kernel_size=5, level=lv).cuda().eval()
        x = torch.randn(n, c, h, h, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
        res = {}
        for env in ("1", "0"):
            os.environ["RCX_CPT16"] = env
            _lib.load().rcx_reload_options()
            with torch.no_grad():
                mod(x); torch.cuda.synchronize()
                tf, _ = time_fresh(lambda: mod(x), x, 30)
                tl, _ = time_fn(lambda: mod(x), 30)
            res[env] = (round(tf * 1e3, 1), round(tl * 1e3, 1))
        print(json.dumps({"N": n, "C": c, "H": h, "tiled16_us(fresh,loop)": res["1"], "lanes_us(fresh,loop)": res["0"]}), flush=True)
