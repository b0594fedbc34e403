"""GPU: the end-to-end example (encode -> index -> dataset_search -> late fusion) keeps running."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_end_to_end_example_runs(capsys):
    path = os.path.join(os.path.dirname(os.path.dirname(__file__)), "examples", "end_to_end.py")
    spec = importlib.util.spec_from_file_location("end_to_end_example", path)
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    module.main()
    out = capsys.readouterr().out
    assert "indexes: ['dpr', 'clip']" in out and "fused run of q0" in out
