"""Minimal stand-in for allennlp.common.registrable.Registrable (AllenNLP 0.9 is not
installable here): `@Base.register(name)` + `Base.by_name(name)`, per base class."""
from collections import defaultdict


class Registrable:
    _registry = defaultdict(dict)

    @classmethod
    def register(cls, name):
        registry = Registrable._registry[cls]

        def deco(sub):
            registry[name] = sub
            return sub
        return deco

    @classmethod
    def by_name(cls, name):
        for base, reg in Registrable._registry.items():
            if issubclass(cls, base) or issubclass(base, cls):
                if name in reg:
                    return reg[name]
        raise KeyError('%s is not a registered %s' % (name, cls.__name__))

    @classmethod
    def list_available(cls):
        return sorted(Registrable._registry[cls])
