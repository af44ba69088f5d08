"""oracle/plonk.py -- TEST INFRASTRUCTURE ONLY (CPU oracle; never imported by the product).

Restatement of the plonky2 1.0.0 PLONK prover and verifier for the recursion layer: what the
reference runs after every segment STARK -- `StarkWrapperCircuit::prove` / `shrink` and `root.circuit.prove`
(evm_arithmetization/src/fixed_recursive_verifier.rs:2146, 3167-3179; circuit configs :69, 3081-3165 =
`CircuitConfig::standard_recursion_config()`: 135 wires, 80 routed, 2 constants, 2 challenges, quotient degree factor
8, FRI rate_bits 3 / cap 4 / 28 queries / 16 PoW bits / arity 4).  The arithmetic lives in the un-vendored crate
plonky2 1.0.0 (Cargo.lock:3702-3705): [EXT] plonk/prover.rs `prove_with_partition_witness`, plonk/vanishing_poly.rs
`eval_vanishing_poly(_base_batch)`, plonk/plonk_common.rs (`ZeroPolyOnCoset`, `reduce_with_powers_multi`),
plonk/permutation_argument.rs + plonk/prover.rs `wires_permutation_partial_products_and_zs`,
gates/selectors.rs, gates/gate.rs `eval_filtered` / `compute_filter`, gates/{arithmetic_base, constant,
public_input, noop}.rs, plonk/proof.rs `OpeningSet`, plonk/circuit_data.rs `get_fri_instance`,
plonk/verifier.rs `verify_with_challenges`.  PARITY UNPINNED: the reference tree holds no golden PLONK proof and
cannot be built here; this file is self-consistent (prover <-> verifier) and the HIP library must equal it word for
word.

Scope of this slice: the permutation argument, the public-input binding, selectors and the four gate types a circuit
of constants and base-field arithmetic needs.  The recursion circuits add Poseidon / PoseidonMds / BaseSum /
RandomAccess / Reducing(+Extension) / ArithmeticExtension / MulExtension / Exponentiation / CosetInterpolation gates:
DESIGN.md (PLONK plan) lists them with their constraint counts; the vanishing-polynomial driver below is already
written against a gate list, not against these four.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List

import numpy as np

from . import stark as S
from . import tape as T

P = S.P
G = S.G                      # F::coset_shift() == MULTIPLICATIVE_GROUP_GENERATOR
UNUSED_SELECTOR = (1 << 32) - 1


# ---- quadratic extension (verifier side only) ---------------------------------------------------------------------
class Ext:
    """F_p[X]/(X^2 - 7) with the operators the vanishing-polynomial code uses (+ - * with Ext / int)."""
    __slots__ = ("a", "b")

    def __init__(self, a, b=0):
        self.a, self.b = a % P, b % P

    @staticmethod
    def of(x):
        return x if isinstance(x, Ext) else Ext(int(x))

    def __add__(self, o):
        o = Ext.of(o)
        return Ext(self.a + o.a, self.b + o.b)
    __radd__ = __add__

    def __sub__(self, o):
        o = Ext.of(o)
        return Ext(self.a - o.a, self.b - o.b)

    def __rsub__(self, o):
        return Ext.of(o) - self

    def __mul__(self, o):
        o = Ext.of(o)
        return Ext(self.a * o.a + 7 * self.b * o.b, self.a * o.b + self.b * o.a)
    __rmul__ = __mul__

    def __neg__(self):
        return Ext(-self.a, -self.b)

    def __eq__(self, o):
        o = Ext.of(o)
        return self.a == o.a and self.b == o.b

    def inverse(self):
        nrm = pow((self.a * self.a - 7 * self.b * self.b) % P, P - 2, P)
        return Ext(self.a * nrm, -self.b * nrm)

    def pow(self, e):
        r, b = Ext(1), self
        while e:
            if e & 1:
                r = r * b
            b = b * b
            e >>= 1
        return r


# ---- gates ([EXT] plonky2 gates/*.rs).  eval_unfiltered(local_constants (selectors removed), local_wires, pi_hash) ---
class NoopGate:
    id, degree, num_constants, num_constraints = "NoopGate", 0, 0, 0
    KIND, PARAM = 0, 0

    def eval_unfiltered(self, consts, wires, pi_hash):
        return []


class ConstantGate:
    """gates/constant.rs: wire i carries constant i."""
    degree = 1
    KIND = 1

    def __init__(self, num_consts=2):
        self.num_consts = self.num_constants = self.num_constraints = self.PARAM = num_consts
        self.id = "ConstantGate { num_consts: %d }" % num_consts

    def eval_unfiltered(self, consts, wires, pi_hash):
        return [consts[i] - wires[i] for i in range(self.num_consts)]


class PublicInputGate:
    """gates/public_input.rs: wires 0..3 equal the hash of the public inputs."""
    id, degree, num_constants, num_constraints = "PublicInputGate", 1, 0, 4
    KIND, PARAM = 2, 0

    def eval_unfiltered(self, consts, wires, pi_hash):
        return [wires[i] - pi_hash[i] for i in range(4)]


class ArithmeticGate:
    """gates/arithmetic_base.rs: num_ops x (output = c0 * multiplicand_0 * multiplicand_1 + c1 * addend);
    wires of op i: 4i (multiplicand_0), 4i+1 (multiplicand_1), 4i+2 (addend), 4i+3 (output)."""
    degree, num_constants = 3, 2
    KIND = 3

    def __init__(self, num_ops=20):                      # new_from_config: num_routed_wires / 4
        self.num_ops = self.num_constraints = self.PARAM = num_ops
        self.id = "ArithmeticGate { num_ops: %d }" % num_ops

    def eval_unfiltered(self, consts, wires, pi_hash):
        c0, c1 = consts[0], consts[1]
        return [wires[4 * i + 3] - (wires[4 * i] * wires[4 * i + 1] * c0 + wires[4 * i + 2] * c1)
                for i in range(self.num_ops)]


# ---- extension-field helpers on component pairs (elements may be ints, tape symbols or Ext: at zeta plonky2 works in
# the `ExtensionAlgebra` -- pairs of extension elements with the same X^2 = 7 rule) ---------------------------------
def _emul(a, b):
    return (a[0] * b[0] + 7 * (a[1] * b[1]), a[0] * b[1] + a[1] * b[0])


def _eadd(a, b):
    return (a[0] + b[0], a[1] + b[1])


def _esub(a, b):
    return (a[0] - b[0], a[1] - b[1])


def _escale(a, k):
    return (a[0] * k, a[1] * k)


class ArithmeticExtensionGate:
    """gates/arithmetic_extension.rs: num_ops x (output = c0 * m0 * m1 + c1 * addend) over F_{p^2}; wires of op i:
    8i.. multiplicand_0, +2 multiplicand_1, +4 addend, +6 output (D = 2 wires each)."""
    degree, num_constants = 3, 2
    KIND = 4

    def __init__(self, num_ops=10):                      # num_routed_wires / (4 D)
        self.num_ops = self.PARAM = num_ops
        self.num_constraints = 2 * num_ops
        self.id = "ArithmeticExtensionGate { num_ops: %d }" % num_ops

    def eval_unfiltered(self, consts, w, pi_hash):
        out = []
        for i in range(self.num_ops):
            m0, m1, ad, o = [(w[8 * i + 2 * k], w[8 * i + 2 * k + 1]) for k in range(4)]
            out += list(_esub(o, _eadd(_escale(_emul(m0, m1), consts[0]), _escale(ad, consts[1]))))
        return out


class MulExtensionGate:
    """gates/multiplication_extension.rs: num_ops x (output = c0 * m0 * m1); wires of op i: 6i.. m0, +2 m1, +4 output."""
    degree, num_constants = 3, 1
    KIND = 5

    def __init__(self, num_ops=13):                      # num_routed_wires / (3 D)
        self.num_ops = self.PARAM = num_ops
        self.num_constraints = 2 * num_ops
        self.id = "MulExtensionGate { num_ops: %d }" % num_ops

    def eval_unfiltered(self, consts, w, pi_hash):
        out = []
        for i in range(self.num_ops):
            m0, m1, o = [(w[6 * i + 2 * k], w[6 * i + 2 * k + 1]) for k in range(3)]
            out += list(_esub(o, _escale(_emul(m0, m1), consts[0])))
        return out


class BaseSumGate:
    """gates/base_sum.rs `BaseSumGate<2>`: wire 0 = sum_i limb_i 2^i, wires 1.. the limbs, each a bit."""
    degree, num_constants = 2, 0
    KIND = 6

    def __init__(self, num_limbs=63):                    # new_from_config: min(log_floor(p - 1, 2), num_routed_wires - 1)
        self.num_limbs = self.PARAM = num_limbs
        self.num_constraints = 1 + num_limbs
        self.id = "BaseSumGate { num_limbs: %d } + Base: 2" % num_limbs

    def eval_unfiltered(self, consts, w, pi_hash):
        limbs = [w[1 + i] for i in range(self.num_limbs)]
        acc = 0
        for l in reversed(limbs):                         # reduce_with_powers(limbs, B)
            acc = acc * 2 + l
        return [acc - w[0]] + [l * (l - 1) for l in limbs]


class ReducingGate:
    """gates/reducing.rs: acc_{i} = acc_{i-1} * alpha + coeff_i (base-field coefficients, extension accumulator).
    wires: output 0..2, alpha 2..4, old_acc 4..6, coeffs 6..6+n, accs (n - 1 pairs; the last accumulator is the output)."""
    degree, num_constants = 2, 0
    KIND = 7

    def __init__(self, num_coeffs=43):
        self.n = self.PARAM = num_coeffs
        self.num_constraints = 2 * num_coeffs
        self.id = "ReducingGate { num_coeffs: %d }" % num_coeffs

    def _acc(self, w, i):
        if i == self.n - 1:
            return (w[0], w[1])
        s = 6 + self.n + 2 * i
        return (w[s], w[s + 1])

    def eval_unfiltered(self, consts, w, pi_hash):
        alpha, acc, out = (w[2], w[3]), (w[4], w[5]), []
        for i in range(self.n):
            t = _emul(acc, alpha)
            out += list(_esub(self._acc(w, i), (t[0] + w[6 + i], t[1])))
            acc = self._acc(w, i)
        return out


class ReducingExtensionGate(ReducingGate):
    """gates/reducing_extension.rs: the same with extension-field coefficients (coeff i at wires 6 + 2i, 7 + 2i)."""
    KIND = 8

    def __init__(self, num_coeffs=32):
        self.n = self.PARAM = num_coeffs
        self.num_constraints = 2 * num_coeffs
        self.id = "ReducingExtensionGate { num_coeffs: %d }" % num_coeffs

    def _acc(self, w, i):
        if i == self.n - 1:
            return (w[0], w[1])
        s = 6 + 2 * self.n + 2 * i
        return (w[s], w[s + 1])

    def eval_unfiltered(self, consts, w, pi_hash):
        alpha, acc, out = (w[2], w[3]), (w[4], w[5]), []
        for i in range(self.n):
            out += list(_esub(self._acc(w, i), _eadd(_emul(acc, alpha), (w[6 + 2 * i], w[7 + 2 * i]))))
            acc = self._acc(w, i)
        return out


class ExponentiationGate:
    """gates/exponentiation.rs: output = base^(sum power_bit_i 2^i), most significant bit first; wires: base 0, power
    bits 1..n, output n + 1, intermediate values n + 2 ..."""
    degree, num_constants = 4, 0
    KIND = 9

    def __init__(self, num_power_bits=66):               # max_power_bits(135, 80) = min(80 - 2, (135 - 2) / 2)
        self.n = self.PARAM = num_power_bits
        self.num_constraints = num_power_bits + 1
        self.id = "ExponentiationGate { num_power_bits: %d }" % num_power_bits

    def eval_unfiltered(self, consts, w, pi_hash):
        n, base, out = self.n, w[0], []
        inter = [w[2 + n + i] for i in range(n)]
        for i in range(n):
            prev = 1 if i == 0 else inter[i - 1] * inter[i - 1]
            bit = w[1 + (n - 1 - i)]                      # power bits are little-endian wires, consumed from the top
            out.append(inter[i] - prev * (bit * base + (1 - bit)))
        return out + [w[1 + n] - inter[n - 1]]


class PoseidonGate:
    """gates/poseidon.rs: one permutation per row.  wires: input 0..12, output 12..24, swap 24, delta 25..29, the S-box
    inputs of full rounds 1..3 (29..65), of the 22 partial rounds (65..87), of the last 4 full rounds (87..135).
    Evaluated with the plain round function: the same constraint polynomials as plonky2's fast partial rounds (see
    oracle/poseidon_table.py: between S-boxes both are the same affine maps)."""
    degree, num_constants, num_constraints = 7, 0, 123
    KIND, PARAM = 10, 0
    id = "PoseidonGate(PhantomData<plonky2_field::goldilocks_field::GoldilocksField>)"

    def eval_unfiltered(self, consts, w, pi_hash):
        from . import poseidon_table as PT
        RC, mds = PT.RC, PT.mds
        swap = w[24]
        out = [swap * (swap - 1)]
        for i in range(4):
            out.append(swap * (w[i + 4] - w[i]) - w[25 + i])
        state = [w[i] + w[25 + i] for i in range(4)] + [w[i + 4] - w[25 + i] for i in range(4)] + [w[i] for i in range(8, 12)]
        sbox = lambda x: x * x * x * x * x * x * x
        rnd = 0
        for r in range(4):
            state = [state[i] + RC[rnd * 12 + i] for i in range(12)]
            if r != 0:
                for i in range(12):
                    sin = w[29 + 12 * (r - 1) + i]
                    out.append(state[i] - sin)
                    state[i] = sin
            state = mds([sbox(x) for x in state])
            rnd += 1
        for r in range(22):
            state = [state[i] + RC[rnd * 12 + i] for i in range(12)]
            sin = w[65 + r]
            out.append(state[0] - sin)
            state = mds([sbox(sin)] + state[1:])
            rnd += 1
        for r in range(4):
            state = [state[i] + RC[rnd * 12 + i] for i in range(12)]
            for i in range(12):
                sin = w[87 + 12 * r + i]
                out.append(state[i] - sin)
                state[i] = sin
            state = mds([sbox(x) for x in state])
            rnd += 1
        return out + [state[i] - w[12 + i] for i in range(12)]


class RandomAccessGate:
    """gates/random_access.rs `RandomAccessGate { bits, num_copies, num_extra_constants }`: per copy the routed wires
    access_index, claimed_element, 2^bits list items; then the extra-constant wires; the (unrouted) index bits after
    all routed wires.  Per copy: the bits are boolean, recompose to the index, and folding the list pairwise by the bits
    (least significant first) leaves the claimed element; the extra constants equal their wires."""
    KIND = 11

    def __init__(self, bits=4, num_copies=4, num_extra_constants=2):       # new_from_config(standard, 4)
        self.bits, self.num_copies, self.num_extra = bits, num_copies, num_extra_constants
        self.PARAM = bits | (num_copies << 8) | (num_extra_constants << 16)
        self.degree = bits + 1
        self.num_constants = num_extra_constants
        self.num_constraints = num_copies * (bits + 2) + num_extra_constants
        self.id = "RandomAccessGate { bits: %d, num_copies: %d, num_extra_constants: %d }" % (bits, num_copies, num_extra_constants)

    def eval_unfiltered(self, consts, w, pi_hash):
        vec = 1 << self.bits
        routed = (2 + vec) * self.num_copies + self.num_extra
        out = []
        for c in range(self.num_copies):
            base = (2 + vec) * c
            index, claimed = w[base], w[base + 1]
            items = [w[base + 2 + i] for i in range(vec)]
            bits = [w[routed + c * self.bits + i] for i in range(self.bits)]
            out += [b * (b - 1) for b in bits]
            rec = 0
            for b in reversed(bits):
                rec = rec + rec + b
            out.append(rec - index)
            for b in bits:
                items = [items[2 * k] + b * (items[2 * k + 1] - items[2 * k]) for k in range(len(items) // 2)]
            out.append(items[0] - claimed)
        return out + [consts[i] - w[(2 + vec) * self.num_copies + i] for i in range(self.num_extra)]


class PoseidonMdsGate:
    """gates/poseidon_mds.rs: outputs = MDS * inputs on twelve F_{p^2} elements (wires 2i, 2i+1 in; 24 + 2i, 25 + 2i out)."""
    id, degree, num_constants, num_constraints = "PoseidonMdsGate(PhantomData<plonky2_field::goldilocks_field::GoldilocksField>)", 1, 0, 24
    KIND, PARAM = 12, 0

    def eval_unfiltered(self, consts, w, pi_hash):
        from . import poseidon_table as PT
        out = []
        ins = [(w[2 * i], w[2 * i + 1]) for i in range(12)]
        for r in range(12):
            acc = _escale(ins[r], PT.MDS_DIAG[r])
            for i in range(12):
                acc = _eadd(acc, _escale(ins[(i + r) % 12], PT.MDS_CIRC[i]))
            out += list(_esub((w[24 + 2 * r], w[25 + 2 * r]), acc))
        return out


def barycentric_weights(points):
    """[EXT] field/src/interpolation.rs `barycentric_weights`: w_i = 1 / prod_{j != i} (x_i - x_j)"""
    out = []
    for i, xi in enumerate(points):
        d = 1
        for j, xj in enumerate(points):
            if i != j:
                d = d * (xi - xj) % P
        out.append(pow(d, P - 2, P))
    return out


class CosetInterpolationGate:
    """gates/coset_interpolation.rs `CosetInterpolationGate { subgroup_bits, degree, barycentric_weights }`: the value at
    `evaluation_point` of the polynomial through 2^subgroup_bits extension values on the coset shift * H, by the
    barycentric formula accumulated in chunks of (degree - 1) points with the running (eval, partial product) exposed as
    intermediate wires.  wires: shift 0; values 1 + 2i; evaluation_point, evaluation_value; then the intermediate evals,
    the intermediate products, and the shifted evaluation point (all D = 2 wide)."""
    KIND, num_constants = 13, 0

    def __init__(self, subgroup_bits=4, max_degree=8):          # FRI arity 16 under max_quotient_degree_factor 8
        self.subgroup_bits = subgroup_bits
        n_points = 1 << subgroup_bits
        n_inter = (n_points - 2) // (max_degree - 1)
        self.degree = (n_points - 2) // (n_inter + 1) + 2        # with_max_degree: the smallest degree with that many chunks
        self.n_points, self.n_inter = n_points, (n_points - 2) // (self.degree - 1)
        self.PARAM = subgroup_bits | (self.degree << 8)
        w = S.root_of_unity(subgroup_bits)
        self.domain = [pow(w, i, P) for i in range(n_points)]
        self.weights = barycentric_weights(self.domain)
        self.num_constraints = 2 * (2 + 2 * self.n_inter)
        self.start_inter = 1 + 2 * n_points + 4
        self.id = "CosetInterpolationGate { subgroup_bits: %d, degree: %d, barycentric_weights: [..] }" % (subgroup_bits, self.degree)

    def _partial(self, lo, hi, values, x, ev, prod):
        """interpolation.rs `partial_interpolate`"""
        for i in range(lo, hi):
            term = _esub(x, (self.domain[i], 0))
            ev = _eadd(_emul(ev, term), _emul(_escale(values[i], self.weights[i]), prod))
            prod = _emul(prod, term)
        return ev, prod

    def eval_unfiltered(self, consts, w, pi_hash):
        npnt, d, ni, si = self.n_points, self.degree, self.n_inter, self.start_inter
        pair = lambda k: (w[k], w[k + 1])
        shift, point, value = w[0], pair(1 + 2 * npnt), pair(3 + 2 * npnt)
        shifted = pair(si + 4 * ni)
        out = list(_esub(point, _escale(shifted, shift)))
        values = [pair(1 + 2 * i) for i in range(npnt)]
        ev, prod = self._partial(0, d, values, shifted, (0, 0), (1, 0))
        for i in range(ni):
            iev, iprod = pair(si + 2 * i), pair(si + 2 * (ni + i))
            out += list(_esub(iev, ev)) + list(_esub(iprod, prod))
            lo = 1 + (d - 1) * (i + 1)
            ev, prod = self._partial(lo, min(lo + d - 1, npnt), values, shifted, iev, iprod)
        return out + list(_esub(value, ev))


@dataclass
class CircuitConfig:
    """CircuitConfig::standard_recursion_config()"""
    num_wires: int = 135
    num_routed_wires: int = 80
    num_constants: int = 2
    num_challenges: int = 2
    max_quotient_degree_factor: int = 8
    rate_bits: int = 3
    cap_height: int = 4
    proof_of_work_bits: int = 16
    num_query_rounds: int = 28
    arity_bits: int = 4
    final_poly_bits: int = 5
    hasher: int = 0


def selector_polynomials(gates, instance_gate_index, max_degree):
    """[EXT] gates/selectors.rs `selector_polynomials`.  gates: sorted by (degree, id); instance_gate_index[row] = index
    into gates.  -> (selector columns [n_sel][n] of ints, selector_indices[gate], groups[(start, end)])"""
    n, num_gates = len(instance_gate_index), len(gates)
    max_gate_degree = gates[-1].degree
    if max_gate_degree + num_gates - 1 <= max_degree:
        return [list(instance_gate_index)], [0] * num_gates, [(0, num_gates)]
    assert max_gate_degree < max_degree, "No gate can be added: all degrees are too high"
    groups, start = [], 0
    while start < num_gates:
        size = 0
        while start + size < num_gates and size + gates[start + size].degree < max_degree:
            size += 1
        groups.append((start, start + size))
        start += size
    group_of = lambda i: next(k for k, (a, b) in enumerate(groups) if a <= i < b)
    sel = [[UNUSED_SELECTOR] * n for _ in groups]
    for row, gi in enumerate(instance_gate_index):
        sel[group_of(gi)][row] = gi
    return sel, [group_of(i) for i in range(num_gates)], groups


def get_unique_coset_shifts(num_shifts):
    """[EXT] plonk_common / field `get_unique_coset_shifts`: g^0 .. g^(num_shifts - 1)."""
    out, x = [], 1
    for _ in range(num_shifts):
        out.append(x)
        x = x * G % P
    return out


@dataclass
class Circuit:
    """What `ProverOnlyCircuitData` + `CommonCircuitData` hold for this prover."""
    config: CircuitConfig
    degree_bits: int
    gates: list
    selector_indices: List[int]
    groups: List[tuple]
    num_selectors: int
    constants: np.ndarray            # [num_constants_total][n]: selectors first, then gate constants
    sigmas: np.ndarray               # [num_routed][n]  (values k_{j'} * w^{i'} of the permuted cell)
    k_is: List[int]
    circuit_digest: List[int]
    quotient_degree_factor: int = 8
    constants_sigmas_commit: dict = field(default=None, repr=False)

    @property
    def n(self): return 1 << self.degree_bits
    @property
    def num_constants_total(self): return self.constants.shape[0]
    @property
    def num_partial_products(self):
        return -(-self.config.num_routed_wires // self.quotient_degree_factor) - 1
    @property
    def num_gate_constraints(self): return max(g.num_constraints for g in self.gates)

    def gate_descriptors(self):
        """(kind, param, selector_index, group_start, group_end) per gate: the zk_plonk_gate records of the C ABI"""
        return [(g.KIND, g.PARAM, self.selector_indices[i], *self.groups[self.selector_indices[i]])
                for i, g in enumerate(self.gates)]


def compute_filter(row, group, s, many_selector):
    """[EXT] gates/gate.rs `compute_filter`"""
    f = 1
    for i in list(range(group[0], group[1])) + ([UNUSED_SELECTOR] if many_selector else []):
        if i != row:
            f = f * (i - s)
    return f


def eval_vanishing_terms(c: Circuit, x, l0_x, local_constants, s_sigmas, local_wires, local_zs, next_zs, partial_products,
                         pi_hash, betas, gammas):
    """[EXT] vanishing_poly.rs: all terms of the vanishing polynomial at one point, in the order they are alpha-combined:
    L_0(x) (Z(x) - 1) per challenge, the partial-product checks per challenge, then the gate constraints.  Generic over
    the element type (tape symbols on the prover side, Ext at zeta on the verifier side)."""
    cfg = c.config
    nr, chunk, num_prods = cfg.num_routed_wires, c.quotient_degree_factor, c.num_partial_products
    z1_terms, pp_terms = [], []
    for i in range(cfg.num_challenges):
        z_x, z_gx = local_zs[i], next_zs[i]
        z1_terms.append(l0_x * (z_x - 1))
        nums = [local_wires[j] + betas[i] * (c.k_is[j] * x) + gammas[i] for j in range(nr)]
        dens = [local_wires[j] + betas[i] * s_sigmas[j] + gammas[i] for j in range(nr)]
        accs = [z_x] + list(partial_products[i * num_prods:(i + 1) * num_prods]) + [z_gx]
        for k in range(0, nr, chunk):                    # check_partial_products
            num_p, den_p = 1, 1
            for v in nums[k:k + chunk]:
                num_p = num_p * v
            for v in dens[k:k + chunk]:
                den_p = den_p * v
            pp_terms.append(accs[k // chunk] * num_p - accs[k // chunk + 1] * den_p)
    # evaluate_gate_constraints: constraints of different gates share slots (at most one filter is non-zero per row)
    gate_terms = [0] * c.num_gate_constraints
    many = c.num_selectors > 1
    for gi, g in enumerate(c.gates):
        si = c.selector_indices[gi]
        filt = compute_filter(gi, c.groups[si], local_constants[si], many)
        for j, v in enumerate(g.eval_unfiltered(local_constants[c.num_selectors:], local_wires, pi_hash)):
            gate_terms[j] = gate_terms[j] + filt * v
    return z1_terms + pp_terms + gate_terms


# ---- a small circuit family for the tests -----------------------------------------------------------------------------
def build_arithmetic_circuit(degree_bits, seed, cfg: CircuitConfig = None, n_public_inputs=3):
    """A valid circuit + witness over {Noop, Constant, PublicInput, Arithmetic} gates: one PublicInputGate row, a few
    ConstantGate rows, rows of 20 multiply-adds whose operands are wired to earlier outputs / constants (so the
    permutation argument carries real copy constraints), Noop padding.  -> (Circuit, wires [135][n] uint64,
    public_inputs).  The builder side of plonky2 (generators, in-circuit hashing of the public inputs) is not restated:
    the prover takes the finished circuit data and witness, which is what is produced here."""
    cfg = cfg or CircuitConfig()
    rng = np.random.default_rng(seed)
    n = 1 << degree_bits
    gates = sorted([NoopGate(), ConstantGate(cfg.num_constants), PublicInputGate(), ArithmeticGate(cfg.num_routed_wires // 4)],
                   key=lambda g: (g.degree, g.id))
    gidx = {type(g): i for i, g in enumerate(gates)}
    rnd = lambda: int(rng.integers(0, P, dtype=np.uint64))
    wires = [[0] * n for _ in range(cfg.num_wires)]
    gate_of_row = [gidx[NoopGate]] * n
    gate_consts = [[0] * n for _ in range(cfg.num_constants)]
    parent = {}

    def find(a):
        while parent.setdefault(a, a) != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    def connect(a, b):
        parent[find(a)] = find(b)
    public_inputs = [rnd() for _ in range(n_public_inputs)]
    gate_of_row[0] = gidx[PublicInputGate]               # wires 0..3 of row 0 are set to the hash by the prover's caller
    n_const_rows = max(1, n // 64)
    pool = []                                            # (cell, value) usable as operands
    for r in range(1, 1 + n_const_rows):
        gate_of_row[r] = gidx[ConstantGate]
        for i in range(cfg.num_constants):
            v = rnd()
            gate_consts[i][r] = v
            wires[i][r] = v
            pool.append(((r, i), v))
    first_arith = 1 + n_const_rows
    n_arith = max(1, (n - first_arith) * 3 // 4)
    for r in range(first_arith, first_arith + n_arith):
        gate_of_row[r] = gidx[ArithmeticGate]
        c0, c1 = rnd(), rnd()
        gate_consts[0][r], gate_consts[1][r] = c0, c1
        for op in range(cfg.num_routed_wires // 4):
            vals = []
            for k in range(3):
                cell = (r, 4 * op + k)
                if rng.random() < 0.7:
                    src, v = pool[int(rng.integers(0, len(pool)))]
                    connect(cell, src)
                else:
                    v = rnd()
                wires[4 * op + k][r] = v
                vals.append(v)
            out = (vals[0] * vals[1] % P * c0 + vals[2] * c1) % P
            wires[4 * op + 3][r] = out
            pool.append(((r, 4 * op + 3), out))
        if len(pool) > 4096:
            pool = pool[-4096:]
    for w in range(cfg.num_routed_wires, cfg.num_wires):  # unrouted wires: free advice
        for r in range(n):
            wires[w][r] = rnd()
    sel, selector_indices, groups = selector_polynomials(gates, gate_of_row, cfg.max_quotient_degree_factor + 1)
    constants = np.array(sel + gate_consts, dtype=np.uint64)
    # sigma: every copy-constraint class becomes one cycle ([EXT] permutation_argument.rs `get_sigma_polys`)
    k_is = get_unique_coset_shifts(cfg.num_routed_wires)
    w = S.root_of_unity(degree_bits)
    subgroup = [1] * n
    for i in range(1, n):
        subgroup[i] = subgroup[i - 1] * w % P
    classes = {}
    for r in range(n):
        for j in range(cfg.num_routed_wires):
            classes.setdefault(find((r, j)), []).append((r, j))
    sigmas = np.zeros((cfg.num_routed_wires, n), dtype=np.uint64)
    for cells in classes.values():
        for a, b in zip(cells, cells[1:] + cells[:1]):
            sigmas[a[1], a[0]] = k_is[b[1]] * subgroup[b[0]] % P
    circ = Circuit(cfg, degree_bits, gates, selector_indices, groups, len(sel), constants, sigmas, k_is,
                   [rnd() for _ in range(4)])
    return circ, np.array(wires, dtype=np.uint64), public_inputs


def build_mixed_circuit(degree_bits, seed, cfg: CircuitConfig = None, n_public_inputs=3):
    """Like `build_arithmetic_circuit`, with every gate kind of this file on a few rows each, valid witness included:
    extension arithmetic, bit decompositions, reducing chains, exponentiations and Poseidon permutations whose operands
    are wired to earlier results.  Eleven gates in three selector groups (degrees 0..7 under max_degree 9)."""
    from . import poseidon_table as PT
    cfg = cfg or CircuitConfig()
    assert (cfg.num_wires, cfg.num_routed_wires) == (135, 80), "the wide gates below assume the standard wire counts"
    rng = np.random.default_rng(seed)
    n = 1 << degree_bits
    gates = sorted([NoopGate(), ConstantGate(cfg.num_constants), PublicInputGate(), ArithmeticGate(20), ArithmeticExtensionGate(10),
                    MulExtensionGate(13), BaseSumGate(63), ReducingGate(43), ReducingExtensionGate(32), ExponentiationGate(66),
                    PoseidonGate(), RandomAccessGate(4, 4, 2), PoseidonMdsGate(), CosetInterpolationGate(4, 8)],
                   key=lambda g: (g.degree, g.id))
    gidx = {type(g): i for i, g in enumerate(gates)}
    rnd = lambda: int(rng.integers(0, P, dtype=np.uint64))
    wires = [[rnd() for _ in range(n)] for _ in range(cfg.num_wires)]      # everything not set below is free advice
    gate_of_row = [gidx[NoopGate]] * n
    gate_consts = [[0] * n for _ in range(cfg.num_constants)]
    parent = {}

    def find(a):
        while parent.setdefault(a, a) != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    pool = []

    def operand(r, col, p_connect=0.6):
        """value for routed cell (r, col): copied from an earlier result (copy constraint) or fresh"""
        if pool and col < cfg.num_routed_wires and rng.random() < p_connect:
            src, v = pool[int(rng.integers(0, len(pool)))]
            parent[find((r, col))] = find(src)
        else:
            v = rnd()
        wires[col][r] = v
        return v

    def result(r, col, v):
        wires[col][r] = v % P
        if col < cfg.num_routed_wires:
            pool.append(((r, col), v % P))
    emul = lambda a, b: ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
    public_inputs = [rnd() for _ in range(n_public_inputs)]
    gate_of_row[0] = gidx[PublicInputGate]
    row = 1
    for _ in range(2):                                                     # constants
        gate_of_row[row] = gidx[ConstantGate]
        for i in range(cfg.num_constants):
            v = rnd()
            gate_consts[i][row] = v
            result(row, i, v)
        row += 1
    budget = n - row
    per_kind = max(1, budget // 13)
    for kind in (ArithmeticGate, ArithmeticExtensionGate, MulExtensionGate, BaseSumGate, ReducingGate, ReducingExtensionGate,
                 ExponentiationGate, PoseidonGate, RandomAccessGate, PoseidonMdsGate, CosetInterpolationGate):
        for _ in range(per_kind):
            if row >= n:
                break
            r = row
            row += 1
            gate_of_row[r] = gidx[kind]
            if kind is ArithmeticGate:
                c0, c1 = rnd(), rnd()
                gate_consts[0][r], gate_consts[1][r] = c0, c1
                for op in range(20):
                    a, b, c = (operand(r, 4 * op + k) for k in range(3))
                    result(r, 4 * op + 3, a * b % P * c0 + c * c1)
            elif kind is ArithmeticExtensionGate:
                c0, c1 = rnd(), rnd()
                gate_consts[0][r], gate_consts[1][r] = c0, c1
                for op in range(10):
                    v = [operand(r, 8 * op + k) for k in range(6)]
                    m = emul((v[0], v[1]), (v[2], v[3]))
                    result(r, 8 * op + 6, m[0] * c0 + v[4] * c1)
                    result(r, 8 * op + 7, m[1] * c0 + v[5] * c1)
            elif kind is MulExtensionGate:
                c0 = rnd()
                gate_consts[0][r] = c0
                for op in range(13):
                    v = [operand(r, 6 * op + k) for k in range(4)]
                    m = emul((v[0], v[1]), (v[2], v[3]))
                    result(r, 6 * op + 4, m[0] * c0)
                    result(r, 6 * op + 5, m[1] * c0)
            elif kind is BaseSumGate:
                val = int(rng.integers(0, 1 << 63, dtype=np.uint64))
                result(r, 0, val)
                for i in range(63):
                    wires[1 + i][r] = (val >> i) & 1
            elif kind in (ReducingGate, ReducingExtensionGate):
                g = gates[gidx[kind]]
                alpha = (operand(r, 2), operand(r, 3))
                acc = (operand(r, 4), operand(r, 5))
                for i in range(g.n):
                    if kind is ReducingGate:
                        co = (operand(r, 6 + i), 0)
                    else:
                        co = (operand(r, 6 + 2 * i), operand(r, 7 + 2 * i))
                    t = emul(acc, alpha)
                    acc = ((t[0] + co[0]) % P, (t[1] + co[1]) % P)
                    if i == g.n - 1:
                        result(r, 0, acc[0])
                        result(r, 1, acc[1])
                    else:
                        st = (6 + g.n if kind is ReducingGate else 6 + 2 * g.n) + 2 * i
                        wires[st][r], wires[st + 1][r] = acc
            elif kind is ExponentiationGate:
                nb = 66
                base = operand(r, 0)
                bits = [int(rng.integers(0, 2)) for _ in range(nb)]
                cur = 1
                for i in range(nb):
                    prev = 1 if i == 0 else cur * cur % P
                    cur = prev * (base if bits[nb - 1 - i] else 1) % P
                    wires[2 + nb + i][r] = cur
                for i in range(nb):
                    wires[1 + i][r] = bits[i]
                result(r, 1 + nb, cur)
            elif kind is RandomAccessGate:
                c0, c1 = rnd(), rnd()
                gate_consts[0][r], gate_consts[1][r] = c0, c1
                for c in range(4):
                    idx = int(rng.integers(0, 16))
                    items = [operand(r, 18 * c + 2 + i) for i in range(16)]
                    wires[18 * c][r] = idx
                    result(r, 18 * c + 1, items[idx])
                    for i in range(4):
                        wires[74 + 4 * c + i][r] = (idx >> i) & 1
                result(r, 72, c0)
                result(r, 73, c1)
            elif kind is CosetInterpolationGate:
                g = gates[gidx[kind]]
                shift = operand(r, 0)
                vals = [(operand(r, 1 + 2 * i), operand(r, 2 + 2 * i)) for i in range(16)]
                point = (operand(r, 33), operand(r, 34))
                si = pow(shift, P - 2, P)
                shifted = (point[0] * si % P, point[1] * si % P)
                wires[g.start_inter + 4 * g.n_inter][r], wires[g.start_inter + 4 * g.n_inter + 1][r] = shifted
                red = lambda e: (e[0] % P, e[1] % P)
                ev, prod = g._partial(0, g.degree, vals, shifted, (0, 0), (1, 0))
                for i in range(g.n_inter):
                    ev, prod = red(ev), red(prod)
                    wires[g.start_inter + 2 * i][r], wires[g.start_inter + 2 * i + 1][r] = ev
                    wires[g.start_inter + 2 * (g.n_inter + i)][r], wires[g.start_inter + 2 * (g.n_inter + i) + 1][r] = prod
                    lo = 1 + (g.degree - 1) * (i + 1)
                    ev, prod = g._partial(lo, min(lo + g.degree - 1, 16), vals, shifted, ev, prod)
                ev = red(ev)
                result(r, 35, ev[0])
                result(r, 36, ev[1])
            elif kind is PoseidonMdsGate:
                ins = [(operand(r, 2 * i), operand(r, 2 * i + 1)) for i in range(12)]
                for rr in range(12):
                    for comp in range(2):
                        v = ins[rr][comp] * PT.MDS_DIAG[rr] + sum(ins[(i + rr) % 12][comp] * PT.MDS_CIRC[i] for i in range(12))
                        result(r, 24 + 2 * rr + comp, v)
            else:                                                           # PoseidonGate
                inp = [operand(r, i) for i in range(12)]
                swap = int(rng.integers(0, 2))
                wires[24][r] = swap
                for i in range(4):
                    wires[25 + i][r] = swap * (inp[i + 4] - inp[i]) % P
                st = [(inp[i] + wires[25 + i][r]) % P for i in range(4)] + [(inp[i + 4] - wires[25 + i][r]) % P for i in range(4)] + inp[8:]
                sb = lambda x: pow(x, 7, P)
                k = 0
                for rr in range(4):
                    st = [(st[i] + PT.RC[k * 12 + i]) % P for i in range(12)]
                    if rr:
                        for i in range(12):
                            wires[29 + 12 * (rr - 1) + i][r] = st[i]
                    st = [x % P for x in PT.mds([sb(x) for x in st])]
                    k += 1
                for rr in range(22):
                    st = [(st[i] + PT.RC[k * 12 + i]) % P for i in range(12)]
                    wires[65 + rr][r] = st[0]
                    st = [x % P for x in PT.mds([sb(st[0])] + st[1:])]
                    k += 1
                for rr in range(4):
                    st = [(st[i] + PT.RC[k * 12 + i]) % P for i in range(12)]
                    for i in range(12):
                        wires[87 + 12 * rr + i][r] = st[i]
                    st = [x % P for x in PT.mds([sb(x) for x in st])]
                    k += 1
                for i in range(12):
                    result(r, 12 + i, st[i])
            if len(pool) > 4096:
                pool = pool[-4096:]
    sel, selector_indices, groups = selector_polynomials(gates, gate_of_row, cfg.max_quotient_degree_factor + 1)
    constants = np.array(sel + gate_consts, dtype=np.uint64)
    k_is = get_unique_coset_shifts(cfg.num_routed_wires)
    sigmas = _sigma_values(degree_bits, cfg.num_routed_wires, k_is, find)
    circ = Circuit(cfg, degree_bits, gates, selector_indices, groups, len(sel), constants, sigmas, k_is, [rnd() for _ in range(4)])
    return circ, np.array(wires, dtype=np.uint64), public_inputs


def _sigma_values(degree_bits, num_routed, k_is, find):
    """every copy-constraint class becomes one cycle ([EXT] permutation_argument.rs `get_sigma_polys`)"""
    n = 1 << degree_bits
    w = S.root_of_unity(degree_bits)
    subgroup = [1] * n
    for i in range(1, n):
        subgroup[i] = subgroup[i - 1] * w % P
    classes = {}
    for r in range(n):
        for j in range(num_routed):
            classes.setdefault(find((r, j)), []).append((r, j))
    sigmas = np.zeros((num_routed, n), dtype=np.uint64)
    for cells in classes.values():
        for a, b in zip(cells, cells[1:] + cells[:1]):
            sigmas[a[1], a[0]] = k_is[b[1]] * subgroup[b[0]] % P
    return sigmas


def set_public_input_wires(o, circ, wires, public_inputs):
    """the PublicInputGate row carries hash_no_pad(public_inputs) in wires 0..3 -> (wires, pi_hash)"""
    h = [int(x) for x in o.poseidon_hash_no_pad(np.array(public_inputs, dtype=np.uint64))]
    row = next(r for r in range(circ.n) if int(circ.constants[0][r]) == next(i for i, g in enumerate(circ.gates)
                                                                            if isinstance(g, PublicInputGate)))
    for i in range(4):
        wires[i, row] = h[i]
    return wires, h


def _fri_cfg(fri_api, cfg: CircuitConfig):
    return fri_api.make_cfg(rate_bits=cfg.rate_bits, cap_height=cfg.cap_height, hasher=cfg.hasher,
                            num_challenges=cfg.num_challenges, pow_bits=cfg.proof_of_work_bits,
                            queries=cfg.num_query_rounds, arity_bits=cfg.arity_bits, final_poly_bits=cfg.final_poly_bits)


def _setup(L):
    if getattr(L, "_plonk_ready", False):
        return
    vp = C.c_void_p
    L.orc_plonk_partial_products.restype = C.c_int
    L.orc_plonk_partial_products.argtypes = [vp, vp, vp, C.c_size_t, C.c_size_t, C.c_uint, C.c_uint64, C.c_uint64, vp]
    L.orc_plonk_quotient_values.restype = C.c_int
    L.orc_plonk_quotient_values.argtypes = [vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t,
                                            vp, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, vp, C.c_size_t, vp]
    L._plonk_ready = True


def commit_circuit(o, circ: Circuit):
    """`constants_sigmas_commitment` (circuit_builder.rs `build`: from_values of constants ++ sigmas, no blinding)"""
    if circ.constants_sigmas_commit is None:
        cs = np.ascontiguousarray(np.concatenate([circ.constants, circ.sigmas]), dtype=np.uint64)
        circ.constants_sigmas_commit = o.commit_values(cs, rate_bits=circ.config.rate_bits,
                                                       cap_height=circ.config.cap_height, hasher=circ.config.hasher)
    return circ.constants_sigmas_commit


def fri_instance(fri_api, circ: Circuit, zeta, g_zeta):
    """[EXT] circuit_data.rs `get_fri_instance`: every polynomial of the four oracles at zeta, the Zs at g * zeta."""
    cfg = circ.config
    n0 = circ.num_constants_total + cfg.num_routed_wires
    n2 = cfg.num_challenges * (1 + circ.num_partial_products)
    n3 = cfg.num_challenges * circ.quotient_degree_factor
    all_polys = [(0, i) for i in range(n0)] + [(1, i) for i in range(cfg.num_wires)] + [(2, i) for i in range(n2)] + \
                [(3, i) for i in range(n3)]
    return fri_api.FriInstance([(zeta, all_polys), (g_zeta, [(2, i) for i in range(cfg.num_challenges)])])


def prove(o, fri_api, circ: Circuit, wires, public_inputs, timing=None):
    """[EXT] plonk/prover.rs `prove_with_partition_witness` from the full witness on.  wires: [num_wires][n] uint64.
    -> dict(wires_cap, zs_pp_cap, quotient_cap, openings (flat ext pairs: to_fri_openings order, then plonk_zs_next),
            fri, public_inputs)"""
    import time
    L = o.lib
    _setup(L)
    cfg = circ.config
    n, db = circ.n, circ.degree_bits
    fcfg = _fri_cfg(fri_api, cfg)
    t0 = time.perf_counter()

    def lap(k):
        nonlocal t0
        if timing is not None:
            t1 = time.perf_counter()
            timing[k] = timing.get(k, 0.0) + t1 - t0
            t0 = t1
    cs_commit = commit_circuit(o, circ)
    lap("constants_sigmas commitment (once per circuit)")
    pi_hash = [int(x) for x in o.poseidon_hash_no_pad(np.array(public_inputs, dtype=np.uint64))]
    wires = np.ascontiguousarray(wires, dtype=np.uint64)
    w_commit = o.commit_values(wires, rate_bits=cfg.rate_bits, cap_height=cfg.cap_height, hasher=cfg.hasher)
    lap("wires commitment")
    ch = fri_api.new_challenger(o, cfg.hasher)
    obs = lambda e: L.orc_challenger_observe(C.byref(ch), np.array(e, dtype=np.uint64), len(e))
    obs(circ.circuit_digest)                              # challenger.observe_hash(circuit_digest)
    obs(pi_hash)                                          # challenger.observe_hash(public_inputs_hash)
    L.orc_challenger_observe_cap(C.byref(ch), w_commit["cap"], w_commit["cap"].shape[0])
    betas = [L.orc_challenger_get(C.byref(ch)) for _ in range(cfg.num_challenges)]
    gammas = [L.orc_challenger_get(C.byref(ch)) for _ in range(cfg.num_challenges)]
    # partial products and Zs: Zs first, then the partial products of challenge 0, 1, ..
    nch = circ.num_partial_products + 1
    k_is = np.array(circ.k_is, dtype=np.uint64)
    zs, pps = [], []
    sig = np.ascontiguousarray(circ.sigmas)
    for b, g in zip(betas, gammas):
        out = np.zeros((nch, n), dtype=np.uint64)
        rc = L.orc_plonk_partial_products(wires.ctypes.data, sig.ctypes.data, k_is.ctypes.data, cfg.num_routed_wires,
                                          circ.quotient_degree_factor, db, b, g, out.ctypes.data)
        assert rc == 0
        zs.append(out[nch - 1])
        pps += [out[k] for k in range(nch - 1)]
    zs_pp = np.ascontiguousarray(np.stack(zs + pps))
    lap("partial products and Zs")
    z_commit = o.commit_values(zs_pp, rate_bits=cfg.rate_bits, cap_height=cfg.cap_height, hasher=cfg.hasher)
    L.orc_challenger_observe_cap(C.byref(ch), z_commit["cap"], z_commit["cap"].shape[0])
    lap("Zs / partial products commitment")
    alphas = [L.orc_challenger_get(C.byref(ch)) for _ in range(cfg.num_challenges)]
    # quotient: trace eval_vanishing_terms once, run it on the coset of size n * 2^quotient_degree_bits
    qdf = circ.quotient_degree_factor
    qdb = (qdf - 1).bit_length()
    assert qdb <= cfg.rate_bits
    C0, C1, C2 = circ.num_constants_total + cfg.num_routed_wires, cfg.num_wires, zs_pp.shape[0]
    tb = T.TapeBuilder(C0 + C1 + 2 * C2 + 2)
    v = tb.inputs
    consts_in, sig_in = v[:circ.num_constants_total], v[circ.num_constants_total:C0]
    wires_in = v[C0:C0 + C1]
    loc, nxt = v[C0 + C1:C0 + C1 + C2], v[C0 + C1 + C2:C0 + C1 + 2 * C2]
    x_in, l0_in = v[-2], v[-1]
    terms = eval_vanishing_terms(circ, x_in, l0_in, consts_in, sig_in, wires_in, loc[:cfg.num_challenges],
                                 nxt[:cfg.num_challenges], loc[cfg.num_challenges:], pi_hash, betas, gammas)
    tp = tb.finish(terms)
    qvals = np.zeros((cfg.num_challenges, n << qdb), dtype=np.uint64)
    al = np.array(alphas, dtype=np.uint64)
    ptrs = (C.c_void_p * cfg.num_challenges)(*[qvals[k].ctypes.data for k in range(cfg.num_challenges)])
    ops = tp.ops if tp.ops.size else np.zeros((1, 3), dtype=np.uint32)
    rc = L.orc_plonk_quotient_values(ops.ctypes.data, len(tp.ops), tp.consts.ctypes.data, len(tp.consts),
                                     tp.outputs.ctypes.data, len(tp.outputs), cs_commit["leaves"].ctypes.data, C0,
                                     w_commit["leaves"].ctypes.data, C1, z_commit["leaves"].ctypes.data, C2, db,
                                     cfg.rate_bits, qdb, al.ctypes.data, len(al), ptrs)
    assert rc == 0
    lap("quotient values")
    chunks = []
    for a in qvals:
        a = np.ascontiguousarray(a)
        L.orc_coset_ifft(a, db + qdb, G)
        # trim_to_len(quotient_degree) (the upper coefficients vanish for a satisfied circuit), then chunks(degree)
        chunks += [a[j * n:(j + 1) * n].copy() for j in range(qdf)]
    qco = np.ascontiguousarray(np.stack(chunks))
    N = n << cfg.rate_bits
    leaves = np.zeros((N, qco.shape[0]), dtype=np.uint64)
    nd = L.orc_merkle_num_digests(db + cfg.rate_bits, cfg.cap_height)
    digests = np.zeros((nd, 4), dtype=np.uint64)
    cap = np.zeros((1 << cfg.cap_height, 4), dtype=np.uint64)
    L.orc_commit_coeffs(qco, qco.shape[0], db, cfg.rate_bits, cfg.cap_height, cfg.hasher, leaves.ctypes.data,
                        digests.ctypes.data, cap.ctypes.data)
    q_commit = dict(coeffs=qco, leaves=leaves, digests=digests, cap=cap)
    L.orc_challenger_observe_cap(C.byref(ch), cap, cap.shape[0])
    lap("quotient commitment")
    zeta = np.zeros(2, dtype=np.uint64)
    L.orc_challenger_get_ext(C.byref(ch), zeta)
    zeta = (int(zeta[0]), int(zeta[1]))
    assert Ext(*zeta).pow(n) != Ext(1), "Opening point is in the subgroup."
    g = S.root_of_unity(db)
    gz = (zeta[0] * g % P, zeta[1] * g % P)
    inst = fri_instance(fri_api, circ, zeta, gz)
    commits = [cs_commit, w_commit, z_commit, q_commit]
    opn, proof = fri_api.oracle_fri_prove(o, fcfg, db, commits, inst, ch)
    lap("openings + FRI")
    return dict(wires_cap=w_commit["cap"], zs_pp_cap=z_commit["cap"], quotient_cap=cap, openings=opn, fri=proof,
                public_inputs=list(public_inputs), zs_pp=zs_pp, quotient_coeffs=qco, betas=betas, gammas=gammas,
                alphas=alphas, zeta=zeta, final_challenge=L.orc_challenger_get(C.byref(ch)))


def verify(o, fri_api, circ: Circuit, proof):
    """[EXT] plonk/verifier.rs `verify_with_challenges` + get_challenges: re-derive the challenges from the transcript,
    check the vanishing-polynomial identity at zeta against the quotient openings, verify the FRI proof."""
    L = o.lib
    cfg = circ.config
    n, db = circ.n, circ.degree_bits
    fcfg = _fri_cfg(fri_api, cfg)
    cs_commit = commit_circuit(o, circ)
    pi_hash = [int(x) for x in o.poseidon_hash_no_pad(np.array(proof["public_inputs"], dtype=np.uint64))]
    ch = fri_api.new_challenger(o, cfg.hasher)
    obs = lambda e: L.orc_challenger_observe(C.byref(ch), np.array(e, dtype=np.uint64), len(e))
    obs(circ.circuit_digest)
    obs(pi_hash)
    cap_of = lambda k: np.ascontiguousarray(proof[k], dtype=np.uint64)
    L.orc_challenger_observe_cap(C.byref(ch), cap_of("wires_cap"), 1 << cfg.cap_height)
    betas = [L.orc_challenger_get(C.byref(ch)) for _ in range(cfg.num_challenges)]
    gammas = [L.orc_challenger_get(C.byref(ch)) for _ in range(cfg.num_challenges)]
    L.orc_challenger_observe_cap(C.byref(ch), cap_of("zs_pp_cap"), 1 << cfg.cap_height)
    alphas = [L.orc_challenger_get(C.byref(ch)) for _ in range(cfg.num_challenges)]
    L.orc_challenger_observe_cap(C.byref(ch), cap_of("quotient_cap"), 1 << cfg.cap_height)
    z = np.zeros(2, dtype=np.uint64)
    L.orc_challenger_get_ext(C.byref(ch), z)
    zeta = Ext(int(z[0]), int(z[1]))
    opn = np.ascontiguousarray(proof["openings"], dtype=np.uint64).reshape(-1, 2)
    E = [Ext(int(a), int(b)) for a, b in opn]
    nc, nr, nw = circ.num_constants_total, cfg.num_routed_wires, cfg.num_wires
    nz, npp, nq = cfg.num_challenges, cfg.num_challenges * circ.num_partial_products, cfg.num_challenges * circ.quotient_degree_factor
    if len(E) != nc + nr + nw + nz + npp + nq + nz:
        return False, "opening set has the wrong size"
    pos = 0

    def take(k):
        nonlocal pos
        r = E[pos:pos + k]
        pos += k
        return r
    consts, sigmas, wires, zs, pps, quot, zs_next = take(nc), take(nr), take(nw), take(nz), take(npp), take(nq), take(nz)
    zeta_n = zeta.pow(n)
    z_h = zeta_n - 1
    l0 = z_h * (Ext(n) * (zeta - 1)).inverse()            # eval_l_0(n, zeta)
    terms = eval_vanishing_terms(circ, zeta, l0, consts, sigmas, wires, zs, zs_next, pps, pi_hash, betas, gammas)
    for i, alpha in enumerate(alphas):
        acc = Ext(0)
        for t in reversed(terms):                         # reduce_with_powers
            acc = acc * alpha + t
        # quotient(zeta) = sum_j zeta^(n j) chunk_j(zeta)  (reduce_with_powers of the chunk openings with zeta^n)
        q = Ext(0)
        for cj in reversed(quot[i * circ.quotient_degree_factor:(i + 1) * circ.quotient_degree_factor]):
            q = q * zeta_n + cj
        if not acc == z_h * q:
            return False, "vanishing polynomial identity fails for challenge %d" % i
    g = S.root_of_unity(db)
    inst = fri_instance(fri_api, circ, (zeta.a, zeta.b), (zeta.a * g % P, zeta.b * g % P))
    caps = [cs_commit["cap"], cap_of("wires_cap"), cap_of("zs_pp_cap"), cap_of("quotient_cap")]
    cols = [nc + nr, nw, nz + npp, nq]
    ok, why = fri_api.oracle_fri_verify(o, fcfg, db, caps, cols, inst, np.ascontiguousarray(opn.reshape(-1)),
                                        np.ascontiguousarray(proof["fri"], dtype=np.uint64), ch)
    return (True, "") if ok else (False, "FRI verification failed (%d)" % why)
