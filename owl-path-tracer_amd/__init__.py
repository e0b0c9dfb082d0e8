"""owl-path-tracer_amd: MI355X-native path-tracing render loop (hot path of jctemp/owl-path-tracer).

Layout:
  csrc/    HIP megakernel + host BVH builder + the C-ABI (libmi355pt.so, see include/mi355pt.h)
  host/    C++ host entry point mirroring path_tracer/Main.cpp (settings.json / <scene>.json / .obj.scene)
  pyhost/  Python glue for tests and bench.py only (ctypes binding, scene ingestion mirror, stand-in scenes,
           pixel-tile sharding for torch.distributed)

The directory name carries a hyphen; import it through `ptamd.load()` at the repo root
(module name `owl_path_tracer_amd`).
"""
__all__ = ["pyhost"]
