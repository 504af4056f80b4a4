# usage (GPU box, under rocprofv3 --kernel-trace --stats): one upload of the 1M-triangle scene with the guide tree built on the device
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import rust_pathtracer_amd as pt
m = pt.Scene("ModelTest").set_seeds(1, 1, 1).set_model_synthetic(1000, 501, 150.0, (0.0, 150.0, 0.0), 100.0).set_image(1920, 1080, 16, 50, (1, 1, 1)).build()
ctx = pt.Context(0)
ctx.set_option("guide_build", 1)
ctx.upload(m)
ctx.upload(m)
