"""YAML config surface (host logic).  Mirrors the reference's `get_config` (src/utils/configs.py:26-42): a YAML file becomes
a frozen attribute-access node.  The reference uses yacs, whose merge `literal_eval`s string values -- so
`FORCE_INFLUENCE_VAL: None` (a YAML *string*) becomes Python `None` and `LR: 1e-4` a float; this
loader reproduces that."""
import ast
import yaml


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        self.__dict__["_frozen"] = False
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) else _literal(v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        if self.__dict__.get("_frozen"):
            raise AttributeError(f"Attempted to set {k} to {v}, but CfgNode is immutable")
        self[k] = v

    def freeze(self):
        self.__dict__["_frozen"] = True
        for v in self.values():
            if isinstance(v, CfgNode):
                v.freeze()

    def clone(self):
        return CfgNode(self)


def _literal(v):
    if isinstance(v, str):
        try:
            return ast.literal_eval(v)
        except (ValueError, SyntaxError):
            return v
    return v


def get_config(path):
    with open(path) as f:
        cfg = CfgNode(yaml.safe_load(f))
    cfg.freeze()
    return cfg
