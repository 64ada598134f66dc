from .api import (estimate_background_noise, estimate_noise, remove_background_noise,  # noqa: F401
                  remove_noise)

__all__ = ["estimate_background_noise", "remove_background_noise", "estimate_noise", "remove_noise"]
