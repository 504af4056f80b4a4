# usage (GPU box): PT_LIBDIR=rust_pathtracer_amd/build/var_phases python scratch/phase_stats.py [scene]  (a -DPT_DEBUG_PHASES build)
# rounds and mean active lanes of each kind of step of the ordered walk's wave loop (trace_ordered_kernel only)
import sys, ctypes as C
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import rust_pathtracer_amd as pt
from rust_pathtracer_amd import api
which = sys.argv[1] if len(sys.argv) > 1 else "SphereField"
if which == "mesh":
    s = pt.Scene("ModelTest").set_seeds(1, 1, 1).set_model_synthetic(1000, 501, 150.0, (0.0, 150.0, 0.0), 100.0).set_image(1920, 1080, 16, 50, (1, 1, 1)).build()
    spp = 16
else:
    s = pt.Scene("SphereField").set_seeds(1, 1, 1).set_image(1920, 1080, 32, 50, (1, 1, 1)).build()
    spp = 32
ctx = pt.Context(0)
for a in sys.argv[2:]:
    k, v = a.split("="); ctx.set_option(k, int(v))
ctx.upload(s)
cam = s.camera(0.0)
w, h = 1920, 1080
out = (C.c_ulonglong * 32)()
api.libptgpu.pt_debug_read_phases(out, 1)
ctx.accum_reset(w, h)
st = ctx.render_accumulate(cam, w, h, (1, 1, 1), 0, spp, 50, 1)
api.libptgpu.pt_debug_read_phases(out, 1)
names = ["refill", "finish+verify", "guide0", "guide1", "guide2", "guide3", "leaf", "literal"]
tot_rounds = 0
print(f"{which}: rays {st.rays}, ms {st.ms_total:.1f}")
for i, nm in enumerate(names):
    r, l = out[2 * i], out[2 * i + 1]
    print(f"  {nm:8s} rounds {r:12d}  mean active lanes {l / max(r, 1):6.2f}  lane-steps per ray {l / st.rays:6.2f}")
g = max(1, sum(out[2 * i + 1] for i in range(2, 6)))
print(f"  guide lane-steps {g}: no child hit {out[16] / g:.3f}, both hit {out[17] / g:.3f}; taken with a hit in hand {out[18] / g:.3f}, of those with no child hit {out[19] / max(out[18], 1):.3f}")
tnames = ["refill", "finish+verify", "guide", "leaf", "literal", "choose", "store"]
tot = sum(out[24 + i] for i in range(7))
print("  wave-time by part of the loop: " + ", ".join(f"{nm} {out[24 + i] / tot:.3f}" for i, nm in enumerate(tnames)) + f"  (total {tot / 1e9:.2f} G wave-cycles)")
