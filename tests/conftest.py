import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")

# The sharded decode cuts every transpose into k1 subsets; left alone, its cost model picks their number per capture and world
# size -- one for the short captures of this suite.  The suite runs with FOUR unless a test says otherwise, so that every sharded
# parity test goes through the subset machinery (tests of the model's own choice remove the variable).
os.environ.setdefault("WFX_SHARD_CHUNKS", "4")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def manifest():
    import json
    with open(os.path.join(GOLDEN, "manifest.json")) as fh:
        return json.load(fh)


def _recipes():
    sys.path.insert(0, GOLDEN)
    try:
        import recipes
    finally:
        sys.path.remove(GOLDEN)
    return recipes


def input_path(case: dict) -> str:
    """Path of a golden case's input wav, regenerated from its recipe when it is not kept in the repository and not there yet.
    A recipe that does not reproduce the manifest's SHA-256 on this host fails THIS test (one case), nothing else."""
    recipes = _recipes()
    try:
        return recipes.ensure_input(GOLDEN, case)
    except recipes.GoldenInputMismatch as e:
        pytest.fail(str(e), pytrace=False)


def input_by_name(name: str) -> str:
    return input_path(next(c for c in golden_cases() if c["name"] == name))


def load_golden(name: str):
    """The stage arrays the REFERENCE produced for a case (tests/golden/recipes.py: load_golden): `np.load`-like, the image rebuilt
    from the golden stream where only its SHA-256 is stored -- and handed out only if it hashes to the reference's."""
    return _recipes().load_golden(GOLDEN, name)


def golden_cases():
    """The manifest's cases (collection time: nothing is generated here; a test asks `input_path(case)` for its wav)."""
    import json
    with open(os.path.join(GOLDEN, "manifest.json")) as fh:
        return json.load(fh)["cases"]
