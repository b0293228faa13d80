"""Process-wide memo for ORACLE results inside one pytest run.

The model-level GPU tests run once per convolution algorithm (tests/conftest.py: conv_algorithm = winograd / direct) against the
same torch-CPU oracle evaluation of the same seeded inputs; the fp64 oracle of the Xception-150 case alone takes ~70 s of host
time.  The oracle does not depend on the algorithm under test, so its outputs are computed once and shared — the driver's
`pytest -m gpu` has a 1 200 s limit and a timeout would read as "every row untested" (Weak #11)."""
_MEMO = {}


def oracle_once(key, fn):
    """fn() evaluated at most once per `key` (a hashable description of model / seeds / shape / dtype); returns its result.
    Results must be treated as read-only by the tests."""
    if key not in _MEMO:
        _MEMO[key] = fn()
    return _MEMO[key]
