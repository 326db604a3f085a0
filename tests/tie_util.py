"""TEST INFRASTRUCTURE: one switch for the tie order of the product (FCD_TIE_*, include/fcd.h) and of the oracle."""
import contextlib

from oracle import oracle

ORDERS = ("pdq178", "stable")


@contextlib.contextmanager
def tie_order(fcd, order):
    """Within the block both the kernels (process default) and the oracle order equal probabilities above 20
    candidates as `order` says: "pdq178" (Rust 1.78's sort_unstable_by) or "stable" (ascending node index)."""
    prev = fcd.tie_order()
    fcd.set_tie_order(order)
    try:
        with oracle.unstable_sort("pdqsort" if order == "pdq178" else "stable"):
            yield
    finally:
        fcd.set_tie_order(prev)
