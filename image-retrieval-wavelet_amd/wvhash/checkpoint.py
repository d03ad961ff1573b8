"""Loading the weights of a reference checkpoint into the wvhash model classes.

Reference: checkpoint dict written by /root/reference/main/engine/chepoint.py:22-45 (keys ``net_state``,
``optimizer_state``, ``scheduler_*_state``, ``scaler_state``, ``epoch``, ``seed``, ``config``, ``score``, ...)
and consumed by evaluate.py:64-71 (``net.load_state_dict(state["net_state"])``, strict).

Files are opened with ``torch.load(..., weights_only=True)`` only: nothing in the file is executed.  A checkpoint
whose ``config`` entry is an OmegaConf object is refused by that loader; re-save it with the config converted to
a plain dict (``OmegaConf.to_container``) or pass just the ``net_state`` tensor dict.
"""
import torch


def read_checkpoint(path, map_location="cpu"):
    try:
        return torch.load(path, map_location=map_location, weights_only=True)
    except Exception as e:  # pickle.UnpicklingError and friends: say what to do instead of falling back
        raise RuntimeError(
            f"{path}: refused by the weights-only loader ({type(e).__name__}: {e}). wvhash never unpickles arbitrary "
            "objects; re-save the checkpoint with plain containers (e.g. config via OmegaConf.to_container).") from e


def net_state_of(state):
    """The model tensors of a checkpoint dict (or the dict itself when it already is a state_dict), with the
    ``module.`` prefix of nn.DataParallel / DDP wrappers removed."""
    sd = state.get("net_state", state) if isinstance(state, dict) else state
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}


def load_net_state(net, path_or_state, strict=True, map_location="cpu"):
    """``net.load_state_dict(state["net_state"])`` like evaluate.py:69; returns the rest of the checkpoint dict
    (epoch, score, ...) for the caller."""
    state = read_checkpoint(path_or_state, map_location) if isinstance(path_or_state, (str, bytes)) or \
        hasattr(path_or_state, "__fspath__") else path_or_state
    net.load_state_dict(net_state_of(state), strict=strict)
    return {k: v for k, v in state.items() if k != "net_state"} if isinstance(state, dict) and "net_state" in state else {}
