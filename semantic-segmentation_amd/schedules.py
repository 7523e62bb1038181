"""Evaluation of the schedule strings of the reference's configuration files without `eval`.

The reference stores schedules as Python lambda source, e.g. "lambda x: 0.99 if x < 40000 else 0.999" or
"lambda x: np.exp(-5.0*(1.0-(min(float(x), 40000.0)/40000.0))**2)*10.0", and evaluates them with `eval`
(src/trainers.py:1054-1055, 1572, 1581, 1945).  Here the source is parsed with `ast` and interpreted over a closed
vocabulary: numbers, the lambda's own parameters, arithmetic / comparison / boolean / conditional expressions, and calls
of a fixed list of numeric functions (builtins min, max, abs, pow, float, int, round; math.* and np.* by name).
Anything else (attribute access on other objects, subscripts, comprehensions, names outside the list) is rejected.
"""
from __future__ import annotations

import ast
import math
import operator
from typing import Callable

import numpy as np

def _py2_div(a, b):
    """`/` as the reference's interpreter (Python 2, no `from __future__ import division`) evaluates it: floor division
    when both operands are integers, true division otherwise."""
    if isinstance(a, (int, np.integer)) and isinstance(b, (int, np.integer)) and not isinstance(a, bool) and not isinstance(b, bool):
        return a // b
    return a / b


_BIN = {ast.Add: operator.add, ast.Sub: operator.sub, ast.Mult: operator.mul, ast.Div: _py2_div,
        ast.FloorDiv: operator.floordiv, ast.Mod: operator.mod, ast.Pow: operator.pow}
_UN = {ast.UAdd: operator.pos, ast.USub: operator.neg, ast.Not: operator.not_}
_CMP = {ast.Lt: operator.lt, ast.LtE: operator.le, ast.Gt: operator.gt, ast.GtE: operator.ge, ast.Eq: operator.eq,
        ast.NotEq: operator.ne}
_FUNCS = {"min": min, "max": max, "abs": abs, "pow": pow, "float": float, "int": int, "round": round}
_MODULE_FUNCS = ("exp", "log", "log10", "sqrt", "power", "pow", "floor", "ceil", "minimum", "maximum", "clip", "cos", "sin",
                 "tanh", "fabs", "abs")
_MODULE_CONSTS = ("pi", "e")


def _module_attr(mod_name: str, attr: str):
    mod = {"np": np, "numpy": np, "math": math}.get(mod_name)
    if mod is None or not hasattr(mod, attr):
        raise ValueError("schedule: unknown name {}.{}".format(mod_name, attr))
    if attr in _MODULE_FUNCS or attr in _MODULE_CONSTS:
        return getattr(mod, attr)
    raise ValueError("schedule: {}.{} is not allowed".format(mod_name, attr))


def _ev(node, env):
    if isinstance(node, ast.Constant):
        if isinstance(node.value, (int, float, bool)):
            return node.value
        raise ValueError("schedule: constant {!r} is not allowed".format(node.value))
    if isinstance(node, ast.Name):
        if node.id in env:
            return env[node.id]
        if node.id in _FUNCS:
            return _FUNCS[node.id]
        if node.id in ("True", "False"):
            return node.id == "True"
        raise ValueError("schedule: unknown name '{}'".format(node.id))
    if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name):
        return _module_attr(node.value.id, node.attr)
    if isinstance(node, ast.BinOp) and type(node.op) in _BIN:
        return _BIN[type(node.op)](_ev(node.left, env), _ev(node.right, env))
    if isinstance(node, ast.UnaryOp) and type(node.op) in _UN:
        return _UN[type(node.op)](_ev(node.operand, env))
    if isinstance(node, ast.BoolOp):
        # Python's and / or: the value of the operand that decides, not a bool
        is_and = isinstance(node.op, ast.And)
        val = None
        for v in node.values:
            val = _ev(v, env)
            if bool(val) != is_and:
                return val
        return val
    if isinstance(node, ast.Compare):
        left = _ev(node.left, env)
        for op, right in zip(node.ops, node.comparators):
            if type(op) not in _CMP:
                raise ValueError("schedule: comparison not allowed")
            r = _ev(right, env)
            if not _CMP[type(op)](left, r):
                return False
            left = r
        return True
    if isinstance(node, ast.IfExp):
        return _ev(node.body, env) if _ev(node.test, env) else _ev(node.orelse, env)
    if isinstance(node, ast.Call) and not node.keywords:
        fn = _ev(node.func, env)
        if not callable(fn):
            raise ValueError("schedule: call of a non-function")
        return fn(*[_ev(a, env) for a in node.args])
    raise ValueError("schedule: expression element {} is not allowed".format(type(node).__name__))


def parse_schedule(spec) -> Callable:
    """A callable from a schedule specification: a callable (returned as is), a number (constant schedule) or the
    source of a lambda over the step / epoch index."""
    if callable(spec):
        return spec
    if isinstance(spec, (int, float)):
        return lambda step, v=float(spec): v
    tree = ast.parse(str(spec).strip(), mode="eval")
    lam = tree.body
    if not isinstance(lam, ast.Lambda) or lam.args.vararg or lam.args.kwarg or lam.args.kwonlyargs or lam.args.defaults:
        raise ValueError("schedule '{}' is not a plain lambda".format(spec))
    names = [a.arg for a in lam.args.args]
    for n in ast.walk(lam.body):       # reject anything outside the vocabulary at parse time, not at the first call
        if isinstance(n, (ast.Lambda, ast.Subscript, ast.ListComp, ast.GeneratorExp, ast.DictComp, ast.SetComp, ast.Starred,
                          ast.Await, ast.Yield, ast.JoinedStr)):
            raise ValueError("schedule: expression element {} is not allowed".format(type(n).__name__))
        if isinstance(n, ast.Attribute):
            if not isinstance(n.value, ast.Name):
                raise ValueError("schedule: nested attribute access is not allowed")
            _module_attr(n.value.id, n.attr)
        if isinstance(n, ast.Name) and n.id not in names and n.id not in _FUNCS and n.id not in ("np", "numpy", "math"):
            raise ValueError("schedule: unknown name '{}'".format(n.id))

    def fn(*args):
        if len(args) != len(names):
            raise TypeError("schedule takes {} argument(s)".format(len(names)))
        return _ev(lam.body, dict(zip(names, args)))
    fn.__name__ = "schedule"
    return fn
