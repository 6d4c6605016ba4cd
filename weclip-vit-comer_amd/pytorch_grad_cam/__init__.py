from .grad_cam import GradCAM  # noqa: F401
