from .gpu_transform import GpuImageTransform, pil_bilinear_tables  # noqa: F401
