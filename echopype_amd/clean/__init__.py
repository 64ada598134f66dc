from .api import (estimate_background_noise, estimate_noise, mask_attenuated_signal,  # noqa: F401
                  mask_impulse_noise, mask_transient_noise, remove_background_noise, remove_noise)

__all__ = ["estimate_background_noise", "remove_background_noise", "estimate_noise", "remove_noise",
           "mask_transient_noise", "mask_impulse_noise", "mask_attenuated_signal"]
