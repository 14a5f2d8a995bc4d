from .synthetic import SyntheticRoom, get_camera_rays, get_dataset  # noqa: F401
