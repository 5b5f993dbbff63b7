"""timberborn_support_solver_amd — MI355X-native SAT solve loop behind
timberborn_support_solver's solver boundary.  See DESIGN.md / INTEGRATION.md."""
from .encoder import (PLATFORMS_DEFAULT, Cnf, Encoding, EncoderError, PlatformLayout, PlatformLimits,
                      ValidationResult, WorldGrid)
from .loop import run_solver, solver_loop, solver_loop_fan, solver_loop_pair, solver_loop_sweep, weight_loop
from .solver import Mi355Sat, SolverError, SolverResult, algorithmic_bytes

__all__ = ["PLATFORMS_DEFAULT", "Cnf", "Encoding", "EncoderError", "PlatformLayout", "PlatformLimits",
           "ValidationResult", "WorldGrid", "run_solver", "solver_loop", "solver_loop_fan", "solver_loop_pair", "solver_loop_sweep", "weight_loop", "Mi355Sat", "SolverError",
           "SolverResult", "algorithmic_bytes"]
