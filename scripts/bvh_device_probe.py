"""Host-built (SAH) against device-built (linear) BVH on the benchmark scene: build
times per mesh and frame time, printed as a table."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as ra                 # noqa: E402
from raytracer_amd import _abi, render     # noqa: E402

blob = ra.load_sdl(os.path.join(ROOT, "assets", "showcase-4k.sdl")).with_config(width=1920, height=1080)
names = ["teapot (4032 triangles)", "tablecloth (10118 triangles)"]
for mode in ("host", "device"):
    os.environ["RTAMD_BVH"] = mode
    scene = render.DeviceScene(blob)
    secs = []
    for m in range(blob.header.num_meshes):
        v = C.c_double()
        _abi.check(_abi.lib().rtamd_scene_mesh_build_seconds(scene._handle, m, C.byref(v)), "build seconds")
        secs.append(v.value)
    p = render.make_params(rng_mode=_abi.RNG_KEYED)
    scene.render(p)
    _, _, st = scene.render(p)
    print(f"{mode}: build " + ", ".join(f"mesh {m} {1e3 * s:.2f} ms" for m, s in enumerate(secs)) +
          f"; 1080p frame {st.ms_total:.2f} ms, {render.total_rays(st) / st.ms_total / 1e3:.0f} Mrays/s")
    scene.close()
