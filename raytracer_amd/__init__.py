"""MI355X render-loop replacement for kaspermeerts/raytracer (host-side Python plumbing).

The product is the C ABI in include/rtamd.h (librtamd.so: SDL/OBJ/PNG front end
in C, HIP kernels for gfx950).  This package only binds it for tests, the
benchmark and multi-GPU launches.
"""
from .scene import SceneBlob, load_sdl, glibc_rand, quantise_rgb8, ppm_write, png_write  # noqa: F401
