"""Operator layer mirroring ``pfb_imaging.operators`` for the measurement-operator hot path.

The structural Protocols are the reference's own contract
(/root/reference/src/pfb_imaging/operators/__init__.py:37-119): Hessians expose ``dot`` /
``hdot`` (allocating style), preconditioners add ``idot``.
"""

from typing import Protocol, runtime_checkable


@runtime_checkable
class Preconditioner(Protocol):
    def dot(self, x): ...

    def hdot(self, x): ...

    def idot(self, x): ...


@runtime_checkable
class LinearOperator(Protocol):
    def dot(self, x): ...

    def hdot(self, x): ...


def _protocol_members(protocol) -> set:
    members = {name for name in getattr(protocol, "__annotations__", {}) if not name.startswith("_")}
    for name, value in vars(protocol).items():
        if not name.startswith("_") and callable(value):
            members.add(name)
    return members


def require_protocol(obj, protocol, arg_name: str) -> None:
    """TypeError naming the Protocol and the missing members (same message shape as the reference)."""
    if isinstance(obj, protocol):
        return
    missing = sorted(m for m in _protocol_members(protocol) if not hasattr(obj, m))
    detail = f"is missing: {', '.join(missing)}" if missing else "does not conform"
    raise TypeError(f"{arg_name} must satisfy the {protocol.__name__} Protocol; {type(obj).__name__} {detail}")
