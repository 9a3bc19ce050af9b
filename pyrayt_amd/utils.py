"""Host-side helpers of the reference's ``pyrayt/utils.py``: plot colours and the lensmaker's
equation.  Neither is on the device path; they exist so that ``RayTracer.show`` and user scripts
written against upstream keep working."""
import numpy as np

# band edges in microns (utils.py:20-96)
_VIOLET, _BLUE, _CYAN, _GREEN, _YELLOW, _RED, _DEEP_RED = 0.38, 0.44, 0.49, 0.51, 0.58, 0.645, 0.75


def wavelength_to_rgb(wavelength, gamma=0.8):
    """(n,3) RGB for wavelengths in microns: linear ramps between the band edges, the two ends
    faded to 30 %, every channel raised to ``gamma`` (``utils.py:5-102``).  Wavelengths outside
    0.38-0.75 um are clipped to the limits."""
    w = np.asarray(wavelength, dtype=float)
    off, full = np.zeros(w.shape), np.ones(w.shape)

    def ramp(numerator, lo, hi):
        return np.abs(numerator / (hi - lo)) ** gamma

    low = np.maximum(w, _VIOLET)
    fade_in = 0.3 + 0.7 * (low - _VIOLET) / (_BLUE - _VIOLET)
    high = np.minimum(w, _DEEP_RED)
    fade_out = 0.3 + 0.7 * (_DEEP_RED - high) / (_DEEP_RED - _RED)
    bands = [
        (w < _BLUE, (np.abs(-(low - _BLUE) / (_BLUE - _VIOLET) * fade_in) ** gamma, off,
                     np.abs(1.0 * fade_in) ** gamma)),
        ((w >= _BLUE) & (w < _CYAN), (off, ramp(w - _BLUE, _BLUE, _CYAN), full)),
        ((w >= _CYAN) & (w < _GREEN), (off, full, ramp(_GREEN - w, _CYAN, _GREEN))),
        ((w >= _GREEN) & (w < _YELLOW), (ramp(w - _GREEN, _GREEN, _YELLOW), full, off)),
        ((w >= _YELLOW) & (w < _RED), (full, ramp(_RED - w, _YELLOW, _RED), off)),
        (w >= _RED, (np.abs(fade_out) ** gamma, off, off)),
    ]
    masks = [mask for mask, _ in bands]
    channels = [np.select(masks, [rgb[k] for _, rgb in bands]) for k in range(3)]
    return np.stack(channels, axis=-1)


def lensmakers_equation(r1, r2, n_lens, thickness):
    """Paraxial focal length of a thick spherical lens (``utils.py:105-118``); r1 > 0 convex,
    r2 < 0 convex."""
    power = (n_lens - 1) * (1 / r1 - 1 / r2 + (n_lens - 1) * thickness / (n_lens * r1 * r2))
    return 1 / power
