"""Attribute-and-item dictionary over the experiment YAML (the reference uses a YACS-style CfgNode,
general_utils/cfgnode.py:36-141; only the access semantics the hot path and the entry points rely on are
provided: nested attribute + item access, in-place mutation, dump())."""
from __future__ import annotations

import yaml


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, CfgNode) else v) for k, v in self.items()}

    def dump(self, **kw):
        return yaml.safe_dump(self.to_dict(), **kw)

    @classmethod
    def load(cls, path):
        with open(path, "r") as f:
            return cls(yaml.safe_load(f))
