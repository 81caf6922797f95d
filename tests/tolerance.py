"""The floating-point bar of the parity tests (BASELINE north star: "embeddings match within 1e-5 rel fp32")."""
import torch


def assert_embeddings_close(got: torch.Tensor, want: torch.Tensor, rtol: float = 1e-5, what: str = "embeddings") -> None:
    """ELEMENT-WISE ``|got - want| <= rtol * |want| + rtol * rms(want)``: 1e-5 relative for every entry, with an absolute floor tied to the
    TYPICAL magnitude of the reference (its root mean square) — not to its largest entry — so that an entry that cancels to ~0 out of
    O(rms) terms is still held to the rounding error of those terms and every other entry to 1e-5 of itself."""
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    assert got.shape == want.shape, f"{what}: shape {tuple(got.shape)} != {tuple(want.shape)}"
    rms = float(want.pow(2).mean().sqrt()) if want.numel() else 0.0
    err = (got - want).abs()
    bound = rtol * want.abs() + rtol * rms
    bad = err > bound
    if bool(bad.any()):
        worst = int((err - bound).argmax())
        raise AssertionError(f"{what}: {int(bad.sum())} of {want.numel()} entries beyond {rtol:g} relative (+ {rtol:g} * rms = {rtol * rms:.3e}); "
                             f"worst: got {got.flatten()[worst]:.9g}, want {want.flatten()[worst]:.9g}, max abs err {float(err.max()):.3e}")


def assert_gradients_close(got: torch.Tensor, want: torch.Tensor, what: str, rtol: float = 1e-5) -> None:
    """Parameter gradients at the same element-wise bar as the embeddings: ``|got - want| <= rtol * |want| + rtol * rms(want)`` per entry.
    A parameter gradient is a sum over all rows of the graph; where a test sums so many fp32 terms that their rounding noise
    (~ sqrt(rows) * 2^-24 of the terms' magnitude, with cancellation) exceeds 1e-5 of the typical entry, the test passes its own ``rtol`` and
    says why — the default is the north star's 1e-5."""
    assert_embeddings_close(got, want, rtol=rtol, what=what)


def gradient_rtol_needed(got: torch.Tensor, want: torch.Tensor) -> float:
    """The smallest ``rtol`` for which :func:`assert_gradients_close` holds: ``max |got - want| / (|want| + rms(want))``."""
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    rms = float(want.pow(2).mean().sqrt()) if want.numel() else 0.0
    return float(((got - want).abs() / (want.abs() + rms + 1e-300)).max()) if want.numel() else 0.0


def assert_gradients_within(got: torch.Tensor, want: torch.Tensor, what: str, bounds: dict, default: float = 1e-5) -> None:
    """:func:`assert_gradients_close` with a STATED per-tensor bound where the default 1e-5 cannot hold: ``bounds`` maps a substring of the
    parameter name to its rtol (first match wins).  The needed rtol is printed (``pytest -s``), so a bound can be re-measured."""
    rtol = next((v for k, v in bounds.items() if k in what), default)
    print(f"[gradient] {what}: needs rtol {gradient_rtol_needed(got, want):.2e} (bound {rtol:.0e})")
    assert_embeddings_close(got, want, rtol=rtol, what=what)
