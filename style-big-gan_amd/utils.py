"""Registry / config helpers (host code).

Mirror of the reference's ``utils.py`` surface that the models and trainer rely on: ``ClassRegistry`` with the
``add_to_registry(name)`` decorator, ``make_dataclass_from_init`` (a dataclass synthesised from an ``__init__``
signature -- reference utils.py:88-118) and the per-registry aggregate dataclasses.  The reference builds these for
OmegaConf structured configs; OmegaConf is optional here (``config.py`` carries a small yaml + dot-list loader with
the same precedence), so the "no default" marker is our own ``MISSING`` which equals omegaconf's ``'???'``.
"""
import dataclasses
import inspect
import typing

MISSING = "???"     # same sentinel string as omegaconf.MISSING


class EasyDict(dict):
    """dict with attribute access and the in-place ``update`` the model constructors use on their kwargs groups."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        del self[name]


def _field_for(name, param):
    if param.default is inspect.Parameter.empty:
        return (name, typing.Any, MISSING)
    if param.default is None:
        return (name, typing.Optional[typing.Any], None)
    default = param.default
    if isinstance(default, (list, dict, set)) or dataclasses.is_dataclass(default) or isinstance(default, EasyDict):
        return (name, typing.Any, dataclasses.field(default_factory=lambda d=default: _clone_default(d)))
    return (name, type(default), dataclasses.field(default=default))


def _clone_default(d):
    if dataclasses.is_dataclass(d) and not isinstance(d, type):
        return dataclasses.replace(d)
    if isinstance(d, EasyDict):
        return EasyDict(d)
    return type(d)(d)


class KwargsBase:
    """Mixin giving synthesised kwargs dataclasses the dict-like protocol the constructors expect
    (``kwargs.update({...})``, ``**kwargs``, attribute assignment)."""

    def update(self, other=None, **kw):
        for k, v in dict(other or {}, **kw).items():
            setattr(self, k, v)

    def keys(self):
        return [k for k in vars(self) if getattr(self, k) is not MISSING or True]

    def __getitem__(self, k):
        return getattr(self, k)

    def __iter__(self):
        return iter(self.keys())

    def items(self):
        return [(k, getattr(self, k)) for k in self.keys()]


class ClassRegistry:
    def __init__(self):
        self.classes = dict()
        self.args = dict()
        self.arg_keys = None

    def __getitem__(self, item):
        return self.classes[item]

    def __contains__(self, item):
        return item in self.classes

    def make_dataclass_from_init(self, func, name, arg_keys):
        params = [(k, v) for k, v in inspect.signature(func).parameters.items() if k not in ("self", "args", "kwargs")]
        fields = [_field_for(k, v) for k, v in params]
        if arg_keys:
            self.arg_keys = arg_keys
            groups = {key: dataclasses.make_dataclass(key, fields, bases=(KwargsBase,)) for key in arg_keys}
            return dataclasses.make_dataclass(
                name, [(k, v, dataclasses.field(default_factory=v)) for k, v in groups.items()], bases=(KwargsBase,))
        return dataclasses.make_dataclass(name, fields, bases=(KwargsBase,))

    def make_dataclass_from_classes(self, name):
        return dataclasses.make_dataclass(
            name, [(k, v, dataclasses.field(default_factory=v)) for k, v in self.classes.items()], bases=(KwargsBase,))

    def make_dataclass_from_args(self, name):
        return dataclasses.make_dataclass(
            name, [(k, v, dataclasses.field(default_factory=v)) for k, v in self.args.items()], bases=(KwargsBase,))

    def add_to_registry(self, name, arg_keys=None):
        def add_class_by_name(cls):
            self.classes[name] = cls
            self.args[name] = self.make_dataclass_from_init(cls.__init__, name, arg_keys)
            return cls
        return add_class_by_name


def closest_power_of_two(n):
    return 1 << (n - 1).bit_length()


def move_to_device(x, device):
    if isinstance(x, (list, tuple)):
        return x.__class__(move_to_device(t, device) for t in x)
    return x.to(device)
