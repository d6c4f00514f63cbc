"""Operator layer mirroring ``pfb_imaging.operators`` for the measurement-operator hot path.

The reference types its operators structurally (/root/reference/src/pfb_imaging/operators/__init__.py:37-119):
a Hessian is anything with ``dot`` / ``hdot`` (allocating style), a preconditioner adds ``idot``, and callers check
arguments with ``require_protocol(obj, Protocol, arg_name)``, which raises ``TypeError`` naming the Protocol.  This module
keeps that contract; at integration the reference's own Protocols apply to these classes unchanged (they are
``runtime_checkable`` and purely structural).
"""

import inspect
from typing import Protocol, runtime_checkable


@runtime_checkable
class LinearOperator(Protocol):
    """``dot(x)`` applies the operator, ``hdot(x)`` its adjoint; both return a new (or internal) array."""

    def dot(self, x): ...

    def hdot(self, x): ...


@runtime_checkable
class Preconditioner(Protocol):
    """A LinearOperator whose (approximate) inverse is available as ``idot(x)``."""

    def dot(self, x): ...

    def hdot(self, x): ...

    def idot(self, x): ...


def _required_methods(protocol):
    """Names a class must provide to conform: the public functions the Protocol class body itself defines."""
    return sorted(name for name, member in inspect.getmembers(protocol, inspect.isfunction)
                  if not name.startswith("_") and name in vars(protocol))


def require_protocol(obj, protocol, arg_name):
    """Raise ``TypeError`` unless ``obj`` structurally conforms to ``protocol``; the message names the argument, the
    Protocol and whatever ``obj`` lacks."""
    lacking = [name for name in _required_methods(protocol) if not callable(getattr(obj, name, None))]
    if not lacking and isinstance(obj, protocol):
        return
    what = ("lacks " + ", ".join(lacking)) if lacking else "does not conform"
    raise TypeError(f"{arg_name}: a {type(obj).__name__} is not a {protocol.__name__} ({what})")
