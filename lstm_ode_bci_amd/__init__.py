"""MI355X-native LSTM-ODE inner loop: drop-in replacements for the reference's
``EnhancedLSTMModel`` / ``Attention`` (04_lstm_model.py), ``CognitiveStateODE``
(05_ode_model.py, 06_lstm_ode_integration.py) and ``LSTMODEIntegration``
(06_lstm_ode_integration.py), backed by hand-written HIP kernels in ``liblob.so``."""
from .model import AblationLSTMModel, Attention, EnhancedLSTMModel          # noqa: F401
from .ode import CognitiveStateODE                        # noqa: F401
from .integration import LSTMODEIntegration               # noqa: F401

__all__ = ["EnhancedLSTMModel", "AblationLSTMModel", "Attention", "CognitiveStateODE", "LSTMODEIntegration"]
