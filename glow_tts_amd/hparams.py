"""Hyper-parameter loading with the reference's semantics (Arg_Parser.py:3-12, Modules.py:9-13): a yaml file
parsed into nested argparse.Namespace objects.  `get_hp()` reads ./Hyper_Parameters.yaml from the current
working directory (exactly what every reference module does at import time) and falls back to the packaged
defaults; `set_hp()` injects a dict / Namespace (tests, bench)."""
import argparse
import os

import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_YAML = os.path.join(_HERE, "Hyper_Parameters.default.yaml")
_hp = None


def Recursive_Parse(args_Dict):
    ns = argparse.Namespace()
    for key, value in args_Dict.items():
        setattr(ns, key, Recursive_Parse(value) if isinstance(value, dict) else value)
    return ns


def load_yaml(path):
    with open(path, encoding="utf-8") as f:
        return yaml.load(f, Loader=yaml.Loader)


def set_hp(hp):
    global _hp
    _hp = Recursive_Parse(hp) if isinstance(hp, dict) else hp
    return _hp


def get_hp():
    global _hp
    if _hp is None:
        path = "Hyper_Parameters.yaml" if os.path.exists("Hyper_Parameters.yaml") else DEFAULT_YAML
        _hp = Recursive_Parse(load_yaml(path))
    return _hp
