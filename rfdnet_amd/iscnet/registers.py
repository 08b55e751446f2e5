"""Name -> class registries with the reference's registry names and keys
(models/registers.py:5-9, net_utils/registry.py:6-47): the YAML `model:` block
selects sub-networks by `method` name through MODULES.get(name)(cfg, optim_spec)
(models/iscnet/modules/network.py:40-47)."""
import inspect


class Registry(object):
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    def __repr__(self):
        return "%s(name=%s, items=%s)" % (type(self).__name__, self._name, list(self._module_dict))

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key, alter_key=None):
        if key in self._module_dict:
            return self._module_dict[key]
        return self._module_dict.get(alter_key, None)

    def register_module(self, cls):
        if not inspect.isclass(cls):
            raise TypeError("module must be a class, but got %s" % type(cls))
        if cls.__name__ in self._module_dict:
            raise KeyError("%s is already registered in %s" % (cls.__name__, self._name))
        self._module_dict[cls.__name__] = cls
        return cls


METHODS = Registry('method')
MODULES = Registry('module')
LOSSES = Registry('loss')
