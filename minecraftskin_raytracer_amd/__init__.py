"""MI355X-native tile-render hot path of MinecraftSkin_Raytracer (HIP kernels behind a C ABI)."""
from .abi import Config, Mesh, Scene, Texture  # noqa: F401
from .skins import synthetic_skin  # noqa: F401
from .api import (  # noqa: F401
    DeviceScene,
    assemble_frame_device,
    ImageWriter,
    MeshBuilder,
    SceneDesc,
    TileRenderer,
    device_count,
    flatten,
    getBuiltinPoses,
    probe_detmath,
    probe_detmath_range,
    probe_mt_uniform,
    quantize_rgba8,
    quantize_rgba8_device,
    render_png,
    trim,
    unpack_rows_device,
)

__all__ = [
    "Config", "Mesh", "Scene", "Texture", "synthetic_skin", "DeviceScene", "MeshBuilder", "SceneDesc",
    "TileRenderer", "device_count", "flatten", "getBuiltinPoses", "probe_detmath", "probe_detmath_range",
    "probe_mt_uniform", "quantize_rgba8", "quantize_rgba8_device", "unpack_rows_device", "ImageWriter", "render_png", "assemble_frame_device", "trim",
]
