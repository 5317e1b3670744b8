"""`is_bearable` really decides the two hints the reference asks about."""
import typing
def is_bearable(obj, hint):
    origin = typing.get_origin(hint)
    args = typing.get_args(hint)
    if origin is tuple:
        if not isinstance(obj, tuple):
            return False
        if len(args) == 2 and args[1] is Ellipsis:
            return all(is_bearable(o, args[0]) for o in obj)
        if len(args) != len(obj):
            return False
        return all(is_bearable(o, a) for o, a in zip(obj, args))
    if hint is int:
        return isinstance(obj, int) and not isinstance(obj, bool)
    if isinstance(hint, type):
        return isinstance(obj, hint)
    raise NotImplementedError(hint)
