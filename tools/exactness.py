"""How close the GPU is to the CPU oracle sample by sample: fraction of samples whose radiance is bit-identical, and
within 1e-9 relative, on four small scenes (run on a GPU box).  The remaining differences come from the device's
sin/cos/exp/log/atan2 against glibc's; +, *, / and sqrt are correctly rounded on both sides and the build does not
contract multiply-adds."""
import numpy as np, tempfile, sys
sys.path.insert(0,'.')
import pathtracing_amd as pt
from pathtracing_amd import scenes
from oracle import binding as ob
pt.init(0)
for name, gen in (("example", lambda d: scenes.cornell_example(d, WIDTH=128, HEIGHT=96)), ("zoo", lambda d: scenes.material_zoo(d, WIDTH=128, HEIGHT=96)), ("mesh", lambda d: scenes.mesh_box(d, n_lon=64, n_lat=33, material="gold", smooth=True, WIDTH=128, HEIGHT=96)), ("cloud", lambda d: scenes.cloud_box(d, WIDTH=128, HEIGHT=96))):
    d = tempfile.mkdtemp(); gen(d)
    hs = pt.load_scene_directory(d)
    job = pt.make_job(0, 128*96, 8, 0, 5)
    g = pt.DeviceScene(hs).trace_samples(job)[0].reshape(-1,3)
    w = ob.OracleScene(hs).trace_samples(job)[0].reshape(-1,3)
    exact = (g == w).all(axis=1).mean()
    close = np.isclose(g, w, rtol=1e-9, atol=1e-12).all(axis=1).mean()
    print(name, "bit-identical %.5f  within 1e-9 %.5f" % (exact, close))
