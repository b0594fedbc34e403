"""Mirror of the two hot-path helpers of ``meerqat.models.utils``: the ``device`` global
(meerqat/models/utils.py:8) and ``prepare_inputs`` (:85-99)."""
import torch

try:
    from transformers.tokenization_utils_base import BatchEncoding
except Exception:  # pragma: no cover
    BatchEncoding = dict

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


def prepare_inputs(data, device=None):
    """Recursively moves every tensor of a nested dict / BatchEncoding / list / tuple to ``device``
    (default: the module-level ``device``); raises TypeError on anything else, like the reference."""
    target = globals()["device"] if device is None else device
    if isinstance(data, (dict, BatchEncoding)):
        return {k: prepare_inputs(v, target) for k, v in data.items()}
    if isinstance(data, (tuple, list)):
        return type(data)(prepare_inputs(v, target) for v in data)
    if isinstance(data, torch.Tensor):
        return data.to(device=target)
    raise TypeError(f"Unexpected type '{type(data)}' for data:\n{data}")
