"""Pair-interaction rules (host-side mirror of ``src/interactions.jl``).

Only what the grid build needs: the kinds, rule construction, the energy ``rule(r)``
(used for the shift at the cutoff, interactions.jl:367-390), rule sums and the
flattening of a force-field column into the C-ABI rule table.  The radial
derivative formulas themselves (``derivativesGrid``, interactions.jl:432-472) are
evaluated by the HIP kernels; this module only validates that the kinds are ones
the reference accepts in a VdW grid.
"""
from __future__ import annotations

import enum
import math
from dataclasses import dataclass, field
from typing import List, Sequence, Union

from .constants import COULOMBIC_CONVERSION_FACTOR


class FF(enum.IntEnum):
    """``@enum InteractionKind`` -- numeric order of interactions.jl:23-33."""
    HardSphere = 0
    CoulombEwaldDirect = 1
    Coulomb = 2
    LennardJones = 3
    Buckingham = 4
    Monomial = 5
    Exponential = 6
    UndefinedInteraction = 7
    NoInteraction = 8


class Mixing(enum.IntEnum):
    """``@enum MixingRule`` interactions.jl:153-159."""
    LorentzBerthelot = 0
    WaldmanHagler = 1
    Geometric = 2
    IgnoreInteraction = 3
    ErrorOnMix = 4


class UndefinedInteractionError(Exception):
    def __str__(self):
        return "Undefined interaction"


class InvalidParameterNumber(Exception):
    pass


class RepeatedRuleKind(Exception):
    pass


@dataclass
class InteractionRule:
    """interactions.jl:232-237"""
    kind: FF
    params: List[float] = field(default_factory=list)
    shift: float = 0.0
    tailcorrection: bool = True

    def sort_key(self):
        # Base.isless(a, b): kind first, then params lexicographically (interactions.jl:255)
        return (int(self.kind), tuple(self.params))

    def same(self, other) -> bool:
        # Base.:(==) compares kind and params only (interactions.jl:254)
        return isinstance(other, InteractionRule) and self.kind == other.kind and list(self.params) == list(other.params)

    def __call__(self, r: float) -> float:
        """Energy in K at distance r in Å (interactions.jl:367-390)."""
        k, p = self.kind, self.params
        if k == FF.LennardJones:
            x6 = (p[1] / r) ** 6
            v = 4 * p[0] * x6 * (x6 - 1)
        elif k == FF.CoulombEwaldDirect:
            v = COULOMBIC_CONVERSION_FACTOR * p[1] * p[2] * math.erfc(p[0] * r) / r
        elif k == FF.Coulomb:
            v = COULOMBIC_CONVERSION_FACTOR * p[0] * p[1] / r
        elif k == FF.HardSphere:
            v = math.inf if r < p[0] + p[1] else 0.0
        elif k == FF.Buckingham:
            v = p[0] * math.exp(-p[1] * r) - p[2] / (r ** 6)
        elif k == FF.NoInteraction:
            v = 0.0
        elif k == FF.Monomial:
            v = p[0] / r ** p[1]
        elif k == FF.Exponential:
            v = p[0] * math.exp(-p[1] * r)
        elif k == FF.UndefinedInteraction:
            raise UndefinedInteractionError()
        else:  # pragma: no cover
            raise AssertionError
        return v - self.shift

    def at_r2(self, r2: float) -> float:
        """Energy in K from the squared distance in Å² (interactions.jl:392-406): the LJ, hard-sphere
        and monomial kinds avoid the square root, the others defer to ``rule(sqrt(r2))``."""
        k, p = self.kind, self.params
        if k == FF.LennardJones:
            x6 = (p[1] ** 2 / r2) ** 3
            return 4 * p[0] * x6 * (x6 - 1) - self.shift
        if k == FF.HardSphere:
            return (math.inf if r2 < (p[0] + p[1]) ** 2 else 0.0) - self.shift
        if k == FF.NoInteraction:
            return 0.0 - self.shift
        if k == FF.Monomial:
            return p[0] / r2 ** (p[1] / 2) - self.shift
        return self(math.sqrt(r2))

    def tail(self, cutoff: float) -> float:
        """``tailcorrection(rule, cutoff)`` interactions.jl:411-431 (K Å³, before the 2π/V factor)."""
        k, p = self.kind, self.params
        if (not self.tailcorrection or k in (FF.NoInteraction, FF.HardSphere, FF.CoulombEwaldDirect) or math.isinf(cutoff)):
            return 0.0
        if k == FF.LennardJones:
            xlj3 = (p[1] / cutoff) ** 3
            xlj9 = xlj3 ** 3
            return (4 / 3) * p[0] * p[1] ** 3 * (xlj9 / 3 - xlj3)
        if k == FF.Buckingham:
            A, B, Cc = p[:3]
            return A * math.exp(-B * cutoff) * (2.0 + B * cutoff * (2.0 + B * cutoff)) / B ** 3 - Cc / (3 * cutoff ** 3)
        if k == FF.Coulomb:
            raise ValueError("Coulomb direct pair interaction cannot have a tail correction: use Ewald summation.")
        if k == FF.UndefinedInteraction:
            raise UndefinedInteractionError()
        raise AssertionError


def make_rule(kind: FF, *args: float) -> InteractionRule:
    """``(ik::FF.InteractionKind)(args...)`` interactions.jl:276-345."""
    n = len(args)
    a = [float(x) for x in args]
    if kind in (FF.LennardJones, FF.Monomial, FF.Exponential):
        if n != 2:
            raise InvalidParameterNumber(kind, n, [2])
        return InteractionRule(kind, a)
    if kind == FF.CoulombEwaldDirect:
        if n == 3:
            r = InteractionRule(kind, [a[0], a[1], a[2]])
        elif n == 2:
            r = InteractionRule(kind, [a[0], a[1], a[1]])
        else:
            raise InvalidParameterNumber(kind, n, [1, 2])
        if r.params[1] == 0.0 or r.params[2] == 0.0:
            return InteractionRule(FF.NoInteraction, [])
        return r
    if kind == FF.Coulomb:
        if n == 2:
            r = InteractionRule(kind, [a[0], a[1]])
        elif n == 1:
            r = InteractionRule(kind, [a[0], a[0]])
        else:
            raise InvalidParameterNumber(kind, n, [1, 2])
        if r.params[0] == 0.0 or r.params[1] == 0.0:
            return InteractionRule(FF.NoInteraction, [])
        return r
    if kind == FF.HardSphere:
        if n == 2:
            return InteractionRule(kind, [a[0], a[1]])
        if n == 1:
            return InteractionRule(kind, [a[0], a[0]])
        raise InvalidParameterNumber(kind, n, [1, 2])
    if kind == FF.Buckingham:
        if n != 3:
            raise InvalidParameterNumber(kind, n, [3])
        return InteractionRule(kind, a)
    if kind in (FF.NoInteraction, FF.UndefinedInteraction):
        if n != 0:
            raise InvalidParameterNumber(kind, n, [0])
        return InteractionRule(kind, [])
    raise AssertionError  # pragma: no cover


def shifted_rule(kind: FF, params: Sequence[float], shift: bool, cutoff: float,
                 tailcorrection: Union[bool, None] = None) -> InteractionRule:
    """``InteractionRule(kind, params, shift::Bool, cutoff, tailcorrection=!shift)``
    interactions.jl:347-365."""
    if tailcorrection is None:
        tailcorrection = not shift
    if shift:
        rule = InteractionRule(kind, list(params), 0.0, False)
        return InteractionRule(kind, list(params), rule(float(cutoff)), tailcorrection)
    return InteractionRule(kind, list(params), 0.0, tailcorrection)


class InteractionRuleSum:
    """interactions.jl:557-583 -- rules sorted by (kind, params), kinds unique."""

    def __init__(self, rules: Sequence[InteractionRule]):
        rules = list(rules)
        if not rules:
            raise ValueError("`InteractionRuleSum(InteractionRule[])` is ill-defined.")
        if len(rules) == 1:
            raise ValueError("Defining `InteractionRuleSum([rule]) is forbidden. Directly use `rule` instead.")
        srules = sorted(rules, key=InteractionRule.sort_key)
        if srules[-1].kind == FF.NoInteraction:
            raise ValueError("Summing any rule with a `FF.NoInteraction` is forbidden.")
        if srules[-1].kind == FF.UndefinedInteraction:
            raise ValueError("Attempting to sum a rule with a `FF.UndefinedInteraction`.")
        kinds = [r.kind for r in srules]
        if len(set(kinds)) != len(kinds):
            raise RepeatedRuleKind(srules)
        self.rules: List[InteractionRule] = srules

    def __call__(self, r: float) -> float:
        ret = 0.0
        for x in self.rules:
            ret += x(r)
        return ret

    def at_r2(self, r2: float) -> float:
        """interactions.jl:589-595 with a squared distance"""
        ret = 0.0
        for x in self.rules:
            ret += x.at_r2(r2)
        return ret

    def tail(self, cutoff: float) -> float:
        """interactions.jl:646-652"""
        ret = 0.0
        for x in self.rules:
            ret += x.tail(cutoff)
        return ret

    def same(self, other) -> bool:
        return (isinstance(other, InteractionRuleSum) and len(self.rules) == len(other.rules)
                and all(a.same(b) for a, b in zip(self.rules, other.rules)))

    def __repr__(self):
        return f"InteractionRuleSum({self.rules!r})"


Rule = Union[InteractionRule, InteractionRuleSum]


def rules_of(rule: Rule) -> List[InteractionRule]:
    return [rule] if isinstance(rule, InteractionRule) else list(rule.rules)


def map_rule(f, rule: Rule) -> Rule:
    """forcefields.jl:172-178"""
    if isinstance(rule, InteractionRule):
        return f(rule)
    return InteractionRuleSum([f(r) for r in rule.rules])


def _sum_one_rule(l: List[InteractionRule], r: InteractionRule) -> List[InteractionRule]:
    """interactions.jl:612-620"""
    if r.kind == FF.NoInteraction:
        return l
    if r.kind == FF.UndefinedInteraction:
        return [r]
    return l + [r]


def sum_rules(r1: Rule, r2: Rule) -> Rule:
    """interactions.jl:621-644"""
    if isinstance(r1, InteractionRuleSum):
        if isinstance(r2, InteractionRuleSum):
            return InteractionRuleSum(r1.rules + r2.rules)
        return InteractionRuleSum(_sum_one_rule(list(r1.rules), r2))
    if isinstance(r2, InteractionRuleSum):
        return InteractionRuleSum(_sum_one_rule(list(r2.rules), r1))
    if r1.kind == FF.NoInteraction:
        return r2
    if r1.kind == FF.UndefinedInteraction:
        return r1
    l = _sum_one_rule([r1], r2)
    return l[0] if len(l) == 1 else InteractionRuleSum(l)


#: kinds on which ``derivativesGrid`` raises (interactions.jl:442-443,462-467)
_VDW_GRID_ERRORS = {
    FF.Coulomb: "Coulomb interactions should not be taken into account in VdW grids.",
    FF.Monomial: "VdW grid not implemented for Monomial",
    FF.Exponential: "VdW grid not implemented for Exponential",
}


def check_vdw_grid_rule(rule: Rule) -> None:
    """Raise what ``derivativesGrid`` would raise the first time it meets the rule."""
    for r in rules_of(rule):
        if r.kind == FF.UndefinedInteraction:
            raise UndefinedInteractionError()
        if r.kind in _VDW_GRID_ERRORS:
            raise RuntimeError(_VDW_GRID_ERRORS[r.kind])
