"""Import shim: `numba` is not installed here; the reference only uses @numba.jit as a speed-up
(whisper/timing.py:57,82), so an identity decorator preserves its semantics exactly."""


def jit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda f: f
