"""ringsnark_amd -- MI355X-native prover hot path for ringSNARK (ring backend + QRP witness
map + encoding inner products).  See DESIGN.md."""
from . import params, r1cs  # noqa: F401
