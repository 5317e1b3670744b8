"""Stand-in for `beartype`: decorator is identity (oracle test infrastructure)."""
def beartype(fn=None, **kw):
    if fn is None:
        return lambda f: f
    return fn
