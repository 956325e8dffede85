"""MI355X-native path-tracing core for Teetlez/Rust-Raytracer: Python host side.

The directory name has a hyphen, so the package is imported through
`__graft_entry__.load_package()` under the module name `rust_raytracer_amd`.
Everything that computes goes through librtcore.so (HIP, gfx950); see _abi.py.
"""
from . import _abi, hostio, objio, parallel, render, scene, scenes  # noqa: F401
from ._abi import RtError, load  # noqa: F401
from .render import Mode, MultiScene, Renderer, Scene, to_rgb  # noqa: F401
from .scene import SceneDescription, dielectric, glossy, lambertian, metal  # noqa: F401
