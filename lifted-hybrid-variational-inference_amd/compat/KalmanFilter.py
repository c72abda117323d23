"""Drop-in alias of the reference module of the same name: re-exports lhvi.kalman."""
from lhvi.kalman import KalmanFilter  # noqa: F401
