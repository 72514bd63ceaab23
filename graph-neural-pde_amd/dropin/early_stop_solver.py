"""Drop-in for the reference's src/early_stop_solver.py (imported by src/GNN_early.py:12)."""
from gnpde_amd.early_stop_solver import EarlyStopInt, EarlyStopRK4, EarlyStopDopri5, SOLVERS  # noqa: F401
