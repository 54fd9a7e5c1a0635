"""Force field = symmetric matrix of pair rules (mirror of ``src/forcefields.jl``)."""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .interactions import (FF, Mixing, InteractionRule, InteractionRuleSum, Rule, make_rule,
                           map_rule, rules_of, shifted_rule, check_vdw_grid_rule)
from .utils import get_atom_name


class DoubleDefinedInteractionRule(Exception):
    pass


class AsymetricSelfInteractionRule(Exception):
    pass


def _mix_rules(A: InteractionRule, B: InteractionRule, mixing: Mixing) -> InteractionRule:
    """forcefields.jl:51-98"""
    if B.sort_key() < A.sort_key():       # A, B = minmax(A, B)
        A, B = B, A
    if B.kind in (FF.NoInteraction, FF.UndefinedInteraction):
        return InteractionRule(B.kind, [])
    if B.kind == FF.HardSphere:
        return make_rule(FF.HardSphere, A.params[0], B.params[0])
    if A.kind == FF.HardSphere:
        return make_rule(FF.HardSphere, A.params[0], 0.0)
    if B.kind in (FF.CoulombEwaldDirect, FF.Coulomb):
        if A.kind != B.kind:
            raise RuntimeError("Cannot mix cutoff and no-cutoff coulomb interactions")
        if B.kind == FF.CoulombEwaldDirect:
            if A.params[0] != B.params[0]:
                raise RuntimeError("Cannot use two Ewald summations with different cutoffs")
            return make_rule(FF.CoulombEwaldDirect, A.params[0], A.params[1], B.params[1])
        return make_rule(FF.Coulomb, A.params[0], B.params[0])
    if A.kind in (FF.CoulombEwaldDirect, FF.Coulomb):
        return make_rule(FF.NoInteraction)
    if B.kind == FF.LennardJones:
        eA, sA = A.params
        eB, sB = B.params
        if mixing == Mixing.LorentzBerthelot:
            return make_rule(FF.LennardJones, math.sqrt(eA * eB), (sA + sB) / 2)
        if mixing == Mixing.WaldmanHagler:
            sA3, sB3 = sA ** 3, sB ** 3
            sAB6 = (sA3 * sA3 + sB3 * sB3) / 2
            return make_rule(FF.LennardJones, math.sqrt(eA * eB) * sA3 * sB3 / sAB6, sAB6 ** (1 / 6))
        if mixing == Mixing.Geometric:
            return make_rule(FF.LennardJones, math.sqrt(eA * eB), math.sqrt(sA * sB))
        raise AssertionError
    if A.kind != B.kind:
        return make_rule(FF.UndefinedInteraction)
    if B.kind == FF.Buckingham:
        return make_rule(FF.Buckingham, math.sqrt(A.params[0] * B.params[0]),
                         math.sqrt(A.params[1] * B.params[1]), math.sqrt(A.params[2] * B.params[2]))
    # Monomial / Exponential mixing is broken in the reference (calls a MixingRule
    # enum value / a misspelt name, forcefields.jl:89-97): it raises there too.
    raise RuntimeError(f"mixing of {B.kind.name} rules raises in the reference")


def reduce_rule_sum(rules: List[InteractionRule]) -> Rule:
    """forcefields.jl:100-127"""
    rules = sorted(rules, key=InteractionRule.sort_key)
    i = len(rules)
    while i > 0 and rules[i - 1].kind == FF.NoInteraction:
        i -= 1
    if i == 0:
        return make_rule(FF.NoInteraction)
    if i == 1:
        return rules[0]
    if rules[i - 1].kind == FF.UndefinedInteraction:
        return make_rule(FF.UndefinedInteraction)
    rules = rules[:i]
    k = 1
    while k <= i and rules[k - 1].kind == FF.HardSphere:
        k += 1
    if k > 1:
        rule0 = rules.pop(0)
        r1, r2 = rule0.params
        for _ in range(2, k):
            r = rules.pop(0)
            r1 = max(r1, r.params[0])
            r2 = max(r2, r.params[1])
        assert r1 == rule0.params[0]
        newhardsphere = make_rule(FF.HardSphere, r1, r2)
        if k > i:
            return newhardsphere
        rules.insert(0, newhardsphere)
    return InteractionRuleSum(rules)


def asymetric_mix_rules(rule: InteractionRule, is_: InteractionRuleSum, mixing: Mixing) -> Rule:
    """forcefields.jl:129-133 (operator precedence of the first line preserved:
    only an UndefinedInteraction returns early)."""
    if rule.kind != FF.NoInteraction and rule.kind == FF.UndefinedInteraction:
        return rule
    return reduce_rule_sum([_mix_rules(rule, r, mixing) for r in is_.rules])


def mix_rules(rulei: Rule, rulej: Rule, mixing: Mixing) -> Rule:
    """forcefields.jl:135-170.  The sum/sum branch keeps the reference's loop
    condition ``i <= n && m <= j`` and its use of ``rulei`` for both operands."""
    if isinstance(rulei, InteractionRule):
        if isinstance(rulej, InteractionRule):
            return _mix_rules(rulei, rulej, mixing)
        return asymetric_mix_rules(rulei, rulej, mixing)
    if isinstance(rulej, InteractionRule):
        return asymetric_mix_rules(rulej, rulei, mixing)
    i = j = 1
    n, m = len(rulei.rules), len(rulej.rules)
    rules: List[InteractionRule] = []
    while i <= n and m <= j:
        A = rulei.rules[i - 1]
        B = rulei.rules[j - 1]
        if A.kind == B.kind:
            rules.append(_mix_rules(A, B, mixing)); i += 1; j += 1
        elif A.kind == FF.HardSphere:
            rules.append(make_rule(FF.HardSphere, A.params[0], 0.0)); i += 1
        elif B.kind == FF.HardSphere:
            rules.append(make_rule(FF.HardSphere, B.params[0], 0.0)); j += 1
        elif A.kind in (FF.Coulomb, FF.CoulombEwaldDirect):
            i += 1
        elif B.kind in (FF.Coulomb, FF.CoulombEwaldDirect):
            j += 1
        else:
            return make_rule(FF.UndefinedInteraction)
    return reduce_rule_sum(rules)


def _rule_equal(a: Rule, b: Rule) -> bool:
    return a.same(b)


class ForceField:
    """forcefields.jl:5-11.  ``interactions[i][j]`` with 0-based indices; ``sdict`` maps a
    species name to its **1-based** identifier, like the reference."""

    def __init__(self, interactions: List[List[Rule]], sdict: Dict[str, int], symbols: List[str],
                 cutoff: float, name: str = "(unnamed)"):
        self.interactions = interactions
        self.sdict = sdict
        self.symbols = symbols
        self.cutoff = float(cutoff)
        self.name = name

    def __getitem__(self, key) -> Rule:
        a, b = key
        i = self.sdict[a] if isinstance(a, str) else int(a)
        j = self.sdict[b] if isinstance(b, str) else int(b)
        return self.interactions[i - 1][j - 1]

    @property
    def nkinds(self) -> int:
        return len(self.interactions)

    def needsvdwgrid(self, atom: str) -> bool:
        """forcefields.jl:306-316"""
        i = self.sdict[get_atom_name(atom)]
        for row in self.interactions:
            for inter in rules_of(row[i - 1]):
                if inter.kind not in (FF.NoInteraction, FF.CoulombEwaldDirect):
                    return True
        return False

    def lj_only_probe(self, atom: str, framework) -> bool:
        """True when the probe ``atom`` meets every framework kind that is present with at most one Lennard-Jones rule
        (NoInteraction / CoulombEwaldDirect count as none): the condition of the multi-probe grid pass (``ceg_plan_create_multi``)."""
        i = self.sdict[get_atom_name(atom)]
        kinds = {self.sdict[get_atom_name(s)] for s in framework.atomic_symbol}
        for k in kinds:
            real = [r for r in rules_of(self.interactions[k - 1][i - 1]) if r.kind not in (FF.NoInteraction, FF.CoulombEwaldDirect)]
            if len(real) > 1 or (real and real[0].kind != FF.LennardJones):
                return False
        return True

    def rule_table(self, probe: int):
        """Flatten column ``probe`` (1-based) into the C-ABI table:
        ``rules`` structured array + ``rule_offset`` (nkinds+1 int32), i.e. what
        ``derivatives_nocutoff(ff, kind_i, probe, d2)`` (forcefields.jl:302-304) dispatches
        on.  Raises the Julia-side errors for kinds ``derivativesGrid`` rejects."""
        from .._abi import RULE_DTYPE
        flat: List[InteractionRule] = []
        offsets = [0]
        for k in range(self.nkinds):
            rule = self.interactions[k][probe - 1]
            for r in rules_of(rule):
                flat.append(r)
            offsets.append(len(flat))
        table = np.zeros(len(flat), dtype=RULE_DTYPE)
        for t, r in enumerate(flat):
            table[t]['kind'] = int(r.kind)
            for q, v in enumerate(r.params[:3]):
                table[t]['p'][q] = v
            table[t]['shift'] = r.shift
        return table, np.asarray(offsets, dtype=np.int32)

    def pair_table(self):
        """Every pair rule flattened for ``ceg_pairs_create``: ``rules`` structured array and
        ``rule_offset`` (nkinds*nkinds + 1 int32), pair (a, b) 0-based at index ``a*nkinds + b``."""
        from .._abi import RULE_DTYPE
        flat: List[InteractionRule] = []
        offsets = [0]
        for a in range(self.nkinds):
            for b in range(self.nkinds):
                flat.extend(rules_of(self.interactions[a][b]))
                offsets.append(len(flat))
        table = np.zeros(len(flat), dtype=RULE_DTYPE)
        for t, r in enumerate(flat):
            table[t]['kind'] = int(r.kind)
            for q, v in enumerate(r.params[:3]):
                table[t]['p'][q] = v
            table[t]['shift'] = r.shift
        return table, np.asarray(offsets, dtype=np.int32)

    def check_vdw_grid(self, probe: int, kinds_present: Iterable[int]) -> None:
        """Mirror the lazy Julia errors: only kinds actually met in the framework raise."""
        for k in sorted(set(int(x) for x in kinds_present)):
            check_vdw_grid_rule(self.interactions[k - 1][probe - 1])


def build_forcefield(input_: Sequence[Tuple[Tuple[str, str], Rule]], mixing: Mixing = Mixing.ErrorOnMix,
                     cutoff: float = 12.0, shift: bool = True, tailcorrection: Optional[bool] = None,
                     sdict: Optional[Dict[str, int]] = None, name: str = "(unnamed)") -> ForceField:
    """``ForceField(input, mixing, cutoff, shift, tailcorrection; sdict, name)``
    forcefields.jl:204-265."""
    if tailcorrection is None:
        tailcorrection = not shift
    cut = float(cutoff)
    if sdict is None:
        allatoms = sorted({a for (a, _b), _ in input_} | {b for (_a, b), _ in input_})
        smap = {s: i + 1 for i, s in enumerate(allatoms)}
    else:
        smap = sdict
    n = len(smap)
    symbols = [""] * n
    for s, i in smap.items():
        symbols[i - 1] = s
    inter: List[List[Optional[Rule]]] = [[None] * n for _ in range(n)]
    done = np.zeros((n, n), dtype=bool)
    for (a, b), rule in input_:
        i, j = smap[a] - 1, smap[b] - 1
        if i == j:
            for r in rules_of(rule):
                ofs = 1 if r.kind == FF.CoulombEwaldDirect else 0
                if (ofs or r.kind in (FF.HardSphere, FF.Coulomb)) and r.params[ofs] != r.params[1 + ofs]:
                    raise AsymetricSelfInteractionRule(a, r)
        if done[i, j] and not _rule_equal(inter[i][j], rule):
            raise DoubleDefinedInteractionRule(a, b, inter[i][j], rule)
        inter[i][j] = inter[j][i] = rule
        done[i, j] = done[j, i] = True
    for i in range(n):
        for j in range(i + 1, n):
            if done[i, j]:
                continue
            if mixing == Mixing.IgnoreInteraction:
                inter[i][j] = inter[j][i] = make_rule(FF.NoInteraction)
                continue
            if mixing == Mixing.ErrorOnMix or not done[i, i] or not done[j, j]:
                inter[i][j] = inter[j][i] = make_rule(FF.UndefinedInteraction)
                continue
            inter[i][j] = inter[j][i] = mix_rules(inter[i][i], inter[j][j], mixing)
    for i in range(n):
        for j in range(i + 1, n):
            new = map_rule(lambda r: shifted_rule(r.kind, r.params, shift, cut, tailcorrection), inter[i][j])
            inter[i][j] = inter[j][i] = new
    return ForceField(inter, smap, symbols, cut, name)  # type: ignore[arg-type]
