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


@pytest.fixture(scope="session", autouse=True)
def golden_inputs_in_place():
    """The golden inputs are not kept in the repository (tests/golden/recipes.py regenerates them, the manifest pins their SHA-256):
    in place before the first test of a session, whichever file it collects."""
    _recipes().ensure_all(GOLDEN)


def load_golden(name: str):
    """The stage arrays the REFERENCE produced for a case (tests/golden/recipes.py: load_golden): `np.load`-like, the image rebuilt
    from the golden stream where only its SHA-256 is stored -- and handed out only if it hashes to the reference's."""
    return _recipes().load_golden(GOLDEN, name)


def golden_cases():
    """The manifest's cases; inputs that are not kept in the repository (tests/golden/recipes.py) are regenerated on first use."""
    import json
    with open(os.path.join(GOLDEN, "manifest.json")) as fh:
        cases = json.load(fh)["cases"]
    if any("recipe" in c for c in cases):
        sys.path.insert(0, GOLDEN)
        try:
            import recipes
        finally:
            sys.path.remove(GOLDEN)
        for c in cases:
            recipes.ensure_input(GOLDEN, c)
    return cases
