"""Stand-in for `einx` (>=0.3.0) - only the elementwise ops the reference uses.
TEST INFRASTRUCTURE ONLY (oracle/): lets the unmodified reference import in a
container without network.  Semantics follow the package's public behaviour:
`op("a, b -> c", x, y)` broadcasts each operand to the output axes by NAME.
"""
import torch

def _parse(pattern, n_in):
    if '->' in pattern:
        lhs, out = pattern.split('->')
        ins = [s.split() for s in lhs.split(',')]
        out = out.split()
    else:
        ins = [s.split() for s in pattern.split(',')]
        # no '->': output = first operand containing every axis (einx: the union, in order of the longest)
        out = max(ins, key=len)
    assert len(ins) == n_in, (pattern, n_in)
    return ins, out

def _align(t, axes, out):
    if not torch.is_tensor(t):
        return t
    if len(axes) == 0:
        return t
    # numeric literal axes like '1' are singleton axes
    names = [a for a in axes]
    assert t.ndim == len(names), (t.shape, names)
    # permute to output order then insert singleton dims
    present = [a for a in out if a in names]
    perm = [names.index(a) for a in present]
    extra = [i for i, a in enumerate(names) if a not in out]
    assert all(t.shape[i] == 1 for i in extra), (names, out)
    t = t.permute(*perm, *extra).reshape([t.shape[i] for i in perm])
    shape = []
    it = iter(t.shape)
    for a in out:
        shape.append(next(it) if a in present else 1)
    return t.reshape(shape)

def _elementwise(fn):
    def op(pattern, *tensors):
        ins, out = _parse(pattern, len(tensors))
        aligned = [_align(t, ax, out) for t, ax in zip(tensors, ins)]
        return fn(*aligned)
    return op

def _as_tensor_like(ref, v):
    return v if torch.is_tensor(v) else torch.as_tensor(v, device=ref.device)

less = _elementwise(lambda a, b: a < b)
greater = _elementwise(lambda a, b: a > b)
greater_equal = _elementwise(lambda a, b: a >= b)
less_equal = _elementwise(lambda a, b: a <= b)
equal = _elementwise(lambda a, b: a == b)
logical_and = _elementwise(lambda a, b: a & b)
logical_or = _elementwise(lambda a, b: a | b)
multiply = _elementwise(lambda a, b: a * b)
add = _elementwise(lambda a, b: a + b)
subtract = _elementwise(lambda a, b: a - b)

def _where(c, a, b):
    if not torch.is_tensor(a) and not torch.is_tensor(b):
        return torch.where(c, torch.as_tensor(a, device=c.device), torch.as_tensor(b, device=c.device))
    if not torch.is_tensor(a):
        a = torch.as_tensor(a, dtype=b.dtype, device=b.device)
    if not torch.is_tensor(b):
        b = torch.as_tensor(b, dtype=a.dtype, device=a.device)
    return torch.where(c, a, b)

where = _elementwise(_where)
