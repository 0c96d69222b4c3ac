"""The oracle-side RK4 drivers live under oracle/ (oracle/rk4_oracle.py: test infrastructure, also timed by bench.py's
cpu_baseline leg of the RK4-step line); the tests keep importing them under this name."""
from oracle.rk4_oracle import *  # noqa: F401,F403
from oracle.rk4_oracle import solve, solve_westervelt, step_sizes  # noqa: F401
