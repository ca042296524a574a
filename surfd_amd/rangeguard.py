"""Range guard of the split-fp16 ("f16x2") precision mode for the product drivers.

The f16x2 kernels (denoiser convs, decoder) clamp operands to the fp16 range (+-65504) and COUNT every wave / workgroup
that had to (surfd_unet_saturation_count, surfd_decoder_saturation_count).  The reference computes in exact fp32
everywhere (models/mdm.py:46 use_fp16=False; AutoEncoder/models/cbndec.py runs under torch's default fp32), so a
checkpoint whose activations leave the fp16 range must not produce a silently different mesh: a stage that clamped is
re-run in the exact-fp32 precision mode — one line on stderr says so — or, with strict=True, raises.

Host logic only (no torch import): examples/generate.py wires it to the native counters; tests/test_host_logic.py drives
it with stubs.
"""
from __future__ import annotations

import sys
from typing import Callable, Tuple, TypeVar

T = TypeVar("T")


class RangeError(RuntimeError):
    """A stage left the fp16 range of the f16x2 mode and the caller asked for --strict."""


def run_guarded(stage: str, run: Callable[[], T], read_counter: Callable[[], int], to_fp32: Callable[[], None],
                strict: bool = False, log: Callable[[str], None] = None) -> Tuple[T, int]:
    """Runs `run()` in the current precision mode; if the stage's saturation counter (read_counter: returns AND resets
    it) is non-zero afterwards, switches the stage's handle to exact fp32 (to_fp32) and runs it again.

    -> (result, clamped): `clamped` = what the counter said after the first attempt (0: the first result stands).
    strict=True raises RangeError instead of re-running.  A non-zero counter after the fp32 run is a bug in the
    library (the exact mode has no range limit) and raises."""
    if log is None:
        log = lambda msg: print(msg, file=sys.stderr, flush=True)      # noqa: E731
    read_counter()                                   # whatever earlier stages left behind is theirs, not this stage's
    out = run()
    clamped = int(read_counter())
    if clamped == 0:
        return out, 0
    msg = (f"[surfd] {stage}: {clamped} workgroup(s) clamped an operand to the fp16 range in the split-fp16 precision mode "
           f"— the result would differ from the reference's fp32 arithmetic")
    if strict:
        raise RangeError(msg + " (--strict: not re-running; use the exact mode, set_precision('fp32'))")
    log(msg + "; re-running this stage in exact fp32")
    to_fp32()
    out = run()
    again = int(read_counter())
    if again:
        raise RuntimeError(f"{stage}: saturation counter reads {again} in the exact-fp32 mode, which has no range limit")
    return out, clamped
