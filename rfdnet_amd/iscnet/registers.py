"""Name -> class tables for the sub-networks the YAML `model:` block selects by `method`
(reference: models/registers.py:5-9, net_utils/registry.py:6-47; used as
`MODULES.get(name)(cfg, optim_spec)`, models/iscnet/modules/network.py:40-47).

A table is a dict subclass keyed by class name: `@TABLE.register_module` adds a class (and
refuses duplicates and non-classes), `TABLE.get(key, alter_key)` falls back to a second key."""


class Registry(dict):
    def __init__(self, name):
        super().__init__()
        self.name = name

    @property
    def module_dict(self):
        return self

    def get(self, key, alter_key=None):              # noqa: A003 (dict.get with a fallback KEY)
        found = dict.get(self, key)
        return found if found is not None else dict.get(self, alter_key)

    def register_module(self, cls):
        if not isinstance(cls, type):
            raise TypeError("only classes can be registered in %r, got %s" % (self.name, type(cls).__name__))
        if cls.__name__ in self:
            raise KeyError("%r already holds a class named %s" % (self.name, cls.__name__))
        self[cls.__name__] = cls
        return cls

    def __repr__(self):
        return "Registry(%s: %s)" % (self.name, ", ".join(sorted(self)))


METHODS = Registry('method')
MODULES = Registry('module')
LOSSES = Registry('loss')
