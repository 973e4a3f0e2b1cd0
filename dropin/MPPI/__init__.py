"""Drop-in package `MPPI`: engine-backed modules here, everything else from the reference checkout."""
from _ditree_fallthrough import fall_through

__path__ = fall_through(__path__, __name__)
