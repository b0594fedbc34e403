"""GPU: Searcher mirror end to end over the HIP index against the reference-minted golden."""
import json

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["one2many", "many2one_max", "cut"])
def test_searcher_hip_index_matches_reference_golden(case):
    from tests.test_searcher_cpu import GOLDEN, build, check
    from viquae_amd.index import MI355XFlatIndex

    def hip_index(art):
        idx = MI355XFlatIndex(string_factory="Flat", metric_type=0)
        idx.add_vectors(art)
        return idx
    golden = json.load(open(GOLDEN))
    check(build(golden, case, hip_index), golden["cases"][case])
