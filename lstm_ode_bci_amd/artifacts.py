"""On-disk artefacts of the reference, read and written in the reference's own formats
(SURVEY.md §8f rank 1):

* ``lstm_attention_model.pt`` -- ``torch.save({'model_state_dict', 'model_config', 'history'})``
  (04_lstm_model.py:921-933); loaded with ``weights_only=False`` + strict ``load_state_dict``
  (06_lstm_ode_integration.py:416-430).
* ``ode_model.pkl`` -- ``pickle.dump({'params', 'model_class'})`` (05_ode_model.py:773-778).
"""
from __future__ import annotations

import os
import pickle

import torch

from .model import EnhancedLSTMModel
from .ode import CognitiveStateODE

LSTM_FILE = "lstm_attention_model.pt"
ODE_FILE = "ode_model.pkl"


def model_config_of(model, input_size, num_classes=2, dropout=0.4, num_heads=4):
    return {"input_size": input_size, "hidden_size": model.hidden_size, "num_layers": model.num_layers,
            "num_classes": num_classes, "dropout": dropout, "bidirectional": model.bidirectional,
            "num_heads": num_heads}


def save_lstm_checkpoint(model, path, model_config, history=None):
    torch.save({"model_state_dict": model.state_dict(), "model_config": dict(model_config),
                "history": history if history is not None else {}}, path)


def load_lstm_checkpoint(path, device="cuda"):
    """-> (model in eval mode on `device`, model_config, history); strict key/shape match."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    cfg = ckpt["model_config"]
    model = EnhancedLSTMModel(input_size=cfg["input_size"], hidden_size=cfg["hidden_size"],
                              num_layers=cfg["num_layers"], num_classes=cfg["num_classes"],
                              dropout=cfg["dropout"], bidirectional=cfg["bidirectional"],
                              num_heads=cfg.get("num_heads", 4))
    model.load_state_dict(ckpt["model_state_dict"], strict=True)
    return model.to(device).eval(), cfg, ckpt.get("history")


def save_ode_model(ode_model, path):
    with open(path, "wb") as f:
        pickle.dump({"params": dict(ode_model.params), "model_class": "CognitiveStateODE"}, f)


def load_ode_model(path):
    with open(path, "rb") as f:
        data = pickle.load(f)
    return CognitiveStateODE(data["params"])


def load_models(models_path, device="cuda"):
    """Mirror of 06_lstm_ode_integration.py:409-440: (lstm_model, ode_model) from a models directory."""
    lstm, _, _ = load_lstm_checkpoint(os.path.join(models_path, LSTM_FILE), device)
    return lstm, load_ode_model(os.path.join(models_path, ODE_FILE))


def load_processed_sequences(path):
    """``load_data`` of 04_lstm_model.py:250-287 / 09:246-262: the arrays of ``processed_sequences.npz``
    (02_preprocessing.py:400-408) as ``(X_train, y_train, X_val, y_val, X_test, y_test)``; when the archive
    has no validation split, 15 % of the training windows are split off (04:266-276, numpy global RNG)."""
    import numpy as np
    data = np.load(path)
    X_train, y_train = data["X_train"], data["y_train"]
    X_test, y_test = data["X_test"], data["y_test"]
    if "X_val" in data.files and len(data["X_val"]) > 0:
        X_val, y_val = data["X_val"], data["y_val"]
    else:
        n_val = int(len(X_train) * 0.15)
        indices = np.random.permutation(len(X_train))
        X_val, y_val = X_train[indices[:n_val]], y_train[indices[:n_val]]
        X_train, y_train = X_train[indices[n_val:]], y_train[indices[n_val:]]
    return X_train, y_train, X_val, y_val, X_test, y_test
