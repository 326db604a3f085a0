"""A SECOND, deliberately naive restatement of the reference's searches (CRF, duplex; r03: beam_search too) -- test
infrastructure.

Written from /root/reference/src/search.rs:38-157 (crf_beam_search) and src/duplex.rs:7-834 (LogSpace, ProbPair,
SecondaryProbs, build / extend / root probs, beam_search, crf_beam_search) WITHOUT consulting oracle/fcd_oracle.c,
in the most literal Python possible: lists of small objects, the reference's own names and statement order, no
rings, no in-place windows, no shared driver between the plain and the CRF variant.  It exists to remove the
single-reader risk of the C oracle's CRF / duplex halves, which the reference's own vectors pin only thinly
(K6, K14, K15): tests/test_naive_crosscheck.py runs both on ~1000 small random cases (wobbly envelopes, beams
wider than the node count, nodes that leave and re-enter the beam, S in {4, 16}) and demands identical results.

Arithmetic: every value is an IEEE binary32 number held in a Python float; `f32()` rounds after EVERY operation
(binary64 +, -, *, / followed by one rounding to binary32 is the correctly rounded binary32 operation).  ln, exp
and ln_1p are evaluated in binary64 and rounded to binary32 -- the "correctly rounded" definition the kernels and
the oracle's FCDO_MATH_CR mode use; `max_mode` is the reference's default `fastexp`
build, whose exp() is identically 0.
"""
import ctypes
import math

NEG_INF = float("-inf")


def f32(x):
    return ctypes.c_float(x).value


class SearchError(Exception):
    pass


RAN_OUT_OF_BEAM = "Ran out of search space (beam_cut_threshold too high)"       # src/lib.rs:46-53
INCOMPARABLE = "Failed to compare values (NaNs in input?)"
INVALID_ENVELOPE = "Invalid envelope values"


# ------------------------------------------------------------------------------------------------
# src/tree.rs: nodes know their parent, label and payload; children are looked up per (node, label)
# ------------------------------------------------------------------------------------------------
ROOT_NODE = -1


class SuffixTree:
    def __init__(self, alphabet_size):
        self.alphabet_size = alphabet_size
        self.nodes = []          # (label, parent, data)
        self.children = {}       # (node, label) -> child

    def label(self, node):
        return None if node == ROOT_NODE else self.nodes[node][0]

    def info(self, node):
        return None if node == ROOT_NODE else (self.nodes[node][0], self.nodes[node][1])  # (label, parent)

    def add_node(self, parent, label, data):
        assert label < self.alphabet_size
        idx = len(self.nodes)
        self.nodes.append([label, parent, data])
        self.children[(parent, label)] = idx
        return idx

    def get_child(self, node, label):
        return self.children.get((node, label))

    def get_data(self, node):
        return None if node == ROOT_NODE else self.nodes[node][2]

    def iter_from(self, node):
        """leaf -> root: (label, data)"""
        while node != ROOT_NODE:
            label, parent, data = self.nodes[node]
            yield label, data
            node = parent


def stable_sort_by_node(beam):
    return sorted(beam, key=lambda x: x.node)      # sort_by_key is stable, so is sorted()


def sort_by_probability_desc(beam, prob):
    """sort_unstable_by(|a, b| b.probability().partial_cmp(a.probability())): descending.  The order of EQUAL
    probabilities is the stable one (Rust's insertion sort up to 20 elements; above)."""
    keyed = [(prob(x), x) for x in beam]
    if len(keyed) >= 2 and any(p != p for p, _ in keyed):
        raise SearchError(INCOMPARABLE)
    return [x for _, x in sorted(keyed, key=lambda t: -t[0])]


# ------------------------------------------------------------------------------------------------
# src/search.rs:38-157
# ------------------------------------------------------------------------------------------------
class Point1:
    def __init__(self, node, state, label_prob, gap_prob):
        self.node, self.state, self.label_prob, self.gap_prob = node, state, label_prob, gap_prob

    def probability(self):
        return f32(self.label_prob + self.gap_prob)


def crf_beam_search(network_output, init_state, alphabet, beam_size, beam_cut_threshold):
    """network_output: [T][S][N] nested lists of binary32 values; -> (sequence, path)"""
    thr = f32(beam_cut_threshold)
    n_state = len(network_output[0])
    n_base = len(network_output[0][0]) - 1
    tree = SuffixTree(n_base)
    best = 0
    for i, v in enumerate(init_state):      # argmax / max: the first maximum
        if v > init_state[best]:
            best = i
    beam = [Point1(ROOT_NODE, best, init_state[best], init_state[0])]
    for idx, probs in enumerate(network_output):
        next_beam = []
        for b in beam:
            pr = probs[b.state]
            if pr[0] > thr:
                next_beam.append(Point1(b.node, b.state, 0.0, f32(f32(b.label_prob + b.gap_prob) * pr[0])))
            for label in range(n_base):
                pr_b = pr[label + 1]
                if pr_b < thr:
                    continue
                child = tree.get_child(b.node, label)
                if child is None:
                    child = tree.add_node(b.node, label, idx)
                next_beam.append(Point1(child, (b.state * n_base) % n_state + label,
                                        f32(f32(b.label_prob + b.gap_prob) * pr_b), 0.0))
        beam = stable_sort_by_node(next_beam)
        merged = []
        for item in beam:
            if merged and merged[-1].node == item.node:
                merged[-1].label_prob = f32(merged[-1].label_prob + item.label_prob)
                merged[-1].gap_prob = f32(merged[-1].gap_prob + item.gap_prob)
            else:
                merged.append(item)
        beam = sort_by_probability_desc(merged, Point1.probability)[:beam_size]
        if not beam:
            raise SearchError(RAN_OUT_OF_BEAM)
        top = beam[0].probability()
        for x in beam:
            x.label_prob = f32(x.label_prob / top)
            x.gap_prob = f32(x.gap_prob / top)
    path, sequence = [], ""
    for label, time in tree.iter_from(beam[0].node):
        path.append(time)
        sequence += alphabet[label + 1]
    path.reverse()
    return sequence[::-1], path


# ------------------------------------------------------------------------------------------------
# src/search.rs:159-301  beam_search (the north-star path; added in r03 so that the oracle's behaviour on special
# posteriors -- NaN, zeros, values above 1 -- has an independent witness as well)
# ------------------------------------------------------------------------------------------------
def beam_search(network_output, alphabet, beam_size, beam_cut_threshold, collapse_repeats):
    """network_output: [T][N] nested lists of binary32 values; -> (sequence, path)"""
    thr = f32(beam_cut_threshold)
    alphabet_size = len(alphabet) - 1
    tree = SuffixTree(alphabet_size)
    beam = [Point1(ROOT_NODE, 0, 0.0, 1.0)]
    for idx, pr in enumerate(network_output):
        next_beam = []
        for b in beam:
            tip_label = tree.label(b.node)
            if pr[0] > thr:
                next_beam.append(Point1(b.node, b.state, 0.0, f32(f32(b.label_prob + b.gap_prob) * pr[0])))
            for label in range(alphabet_size):
                pr_b = pr[label + 1]
                if pr_b < thr:
                    continue
                if collapse_repeats and label == tip_label:
                    next_beam.append(Point1(b.node, b.state, f32(b.label_prob * pr_b), 0.0))
                    new_node = tree.get_child(b.node, label)
                    if new_node is None and b.gap_prob > 0.0:
                        new_node = tree.add_node(b.node, label, idx)
                    if new_node is not None:
                        next_beam.append(Point1(new_node, b.state, f32(b.gap_prob * pr_b), 0.0))
                else:
                    new_node = tree.get_child(b.node, label)
                    if new_node is None:
                        new_node = tree.add_node(b.node, label, idx)
                    next_beam.append(Point1(new_node, b.state, f32(f32(b.label_prob + b.gap_prob) * pr_b), 0.0))
        beam = stable_sort_by_node(next_beam)
        merged = []
        for item in beam:
            if merged and merged[-1].node == item.node:
                merged[-1].label_prob = f32(merged[-1].label_prob + item.label_prob)
                merged[-1].gap_prob = f32(merged[-1].gap_prob + item.gap_prob)
            else:
                merged.append(item)
        beam = sort_by_probability_desc(merged, Point1.probability)[:beam_size]
        if not beam:
            raise SearchError(RAN_OUT_OF_BEAM)
        top = beam[0].probability()
        for x in beam:
            x.label_prob = f32_div(x.label_prob, top)
            x.gap_prob = f32_div(x.gap_prob, top)
    path, tokens = [], []
    for label, time in tree.iter_from(beam[0].node):
        path.append(time)
        tokens.append(alphabet[label + 1])
    path.reverse()
    tokens.reverse()
    return "".join(tokens), path


def f32_div(a, b):
    """binary32 division with IEEE results for a zero or NaN divisor (Python raises on x / 0.0)"""
    if b != b or a != a:
        return float("nan")
    if b == 0.0:
        if a == 0.0:
            return float("nan")
        return math.copysign(float("inf"), a) * math.copysign(1.0, b)
    return f32(a / b)


# ------------------------------------------------------------------------------------------------
# src/duplex.rs:7-80  LogSpace
# ------------------------------------------------------------------------------------------------
class Log:
    def __init__(self, max_mode):
        self.max_mode = max_mode

    @staticmethod
    def new(x):
        if x != x:
            return x
        if x < 0.0:
            return float("nan")
        return NEG_INF if x == 0.0 else f32(math.log(x))

    def exp(self, a):
        if self.max_mode:
            return 0.0               # src/fastexp.rs: the `fastexp` feature's exp() is identically 0
        if a != a:
            return a
        try:
            return f32(math.exp(a))
        except OverflowError:
            return float("inf")

    def add(self, a, b):
        def add_internal(big, small):
            if small == NEG_INF:
                return big
            e = self.exp(f32(small - big))
            return f32(big + (e if e != e else f32(math.log1p(e))))
        if a <= b:
            return add_internal(b, a)
        return add_internal(a, b)

    @staticmethod
    def mul(a, b):
        return f32(a + b)

    @staticmethod
    def max(a, b):
        return b if a < b else a


class Pair:
    def __init__(self, label=NEG_INF, gap=NEG_INF):
        self.label, self.gap = label, gap


class Secondary:
    def __init__(self, offset):
        self.offset, self.probs, self.max_prob = offset, [], NEG_INF

    def get(self, at):
        index = at - self.offset
        if index < 0 or index >= len(self.probs):
            return Pair()
        return self.probs[index]

    def discard_until(self, keep_from):
        if keep_from > self.offset:
            first_index = keep_from - self.offset
            self.probs = self.probs[first_index:] if first_index < len(self.probs) else []
            self.offset = keep_from

    def end(self):
        return self.offset + len(self.probs)


class Point2:
    def __init__(self, node, state, prob_1, prob_2_max):
        self.node, self.state, self.prob_1, self.prob_2_max = node, state, prob_1, prob_2_max


class Duplex:
    """One object per search: holds the log-space arithmetic flavour."""

    def __init__(self, max_mode):
        self.L = Log(max_mode)

    def pair_probability(self, p):
        return self.L.add(p.label, p.gap)

    def point_probability(self, x):
        return self.L.mul(self.pair_probability(x.prob_1), x.prob_2_max)

    def update_max(self, s, lower_bound, upper_bound):
        assert lower_bound <= upper_bound
        n = len(s.probs)
        begin = min(max(lower_bound - s.offset, 0), n)
        end = min(max(upper_bound - s.offset, begin), n)
        m = NEG_INF
        for p in s.probs[begin:end]:
            m = self.L.max(m, self.pair_probability(p))
        s.max_prob = m

    # :212-249 / :251-289
    def build(self, rows, parent_probs, label, is_repeat, lower_bound, upper_bound):
        assert lower_bound < upper_bound
        s = Secondary(lower_bound)
        last = Pair()
        for idx in range(lower_bound, upper_bound):
            lp = rows(idx)
            gap_prob = self.L.mul(self.pair_probability(last), lp[0])
            prev = parent_probs.get(idx - 1)
            x = prev.gap if is_repeat else self.pair_probability(prev)
            label_prob = self.L.mul(lp[label + 1], self.L.add(last.label, x))
            last = Pair(label_prob, gap_prob)
            s.probs.append(last)
            s.max_prob = self.L.max(s.max_prob, self.pair_probability(last))
        return s

    # :291-336 / :338-387
    def extend(self, s, rows, parent_probs, label, is_repeat, lower_bound, upper_bound):
        assert lower_bound <= upper_bound
        if lower_bound > s.offset:
            s.discard_until(lower_bound - 1)
            if not s.probs:
                s.offset = lower_bound
            self.update_max(s, lower_bound, upper_bound)
        current_end = s.end()
        assert 0 <= current_end < upper_bound
        last = s.probs[-1] if s.probs else Pair()
        for idx in range(current_end, upper_bound):
            lp = rows(idx)
            gap_prob = self.L.mul(self.pair_probability(last), lp[0])
            prev = parent_probs.get(idx - 1)
            x = prev.gap if is_repeat else self.pair_probability(prev)
            label_prob = self.L.mul(lp[label + 1], self.L.add(last.label, x))
            last = Pair(label_prob, gap_prob)
            s.probs.append(last)
            s.max_prob = self.L.max(s.max_prob, self.pair_probability(last))

    def merge_and_prune(self, next_beam, tree, beam_size):
        beam = stable_sort_by_node(next_beam)
        merged = []
        for item in beam:
            if merged and merged[-1].node == item.node:
                merged[-1].prob_1 = Pair(self.L.add(merged[-1].prob_1.label, item.prob_1.label),
                                         self.L.add(merged[-1].prob_1.gap, item.prob_1.gap))
            else:
                merged.append(item)
        for item in merged:
            data = tree.get_data(item.node)
            if data is not None:
                item.prob_2_max = data.max_prob
        beam = sort_by_probability_desc(merged, self.point_probability)[:beam_size]
        if not beam:
            raise SearchError(RAN_OUT_OF_BEAM)
        return beam

    # :443-650
    def beam_search(self, net1_real, net2_real, alphabet, envelope, beam_size, thr_real, collapse_repeats):
        net1 = [[Log.new(v) for v in row] for row in net1_real]
        net2 = [[Log.new(v) for v in row] for row in net2_real]
        thr = Log.new(f32(thr_real))
        alphabet_size = len(alphabet) - 1
        tree = SuffixTree(alphabet_size)
        beam = [Point2(ROOT_NODE, 0, Pair(NEG_INF, 0.0), 0.0)]
        # root_probs (:389-409)
        root = Secondary(-1)
        root.max_prob = 0.0
        cur = 0.0
        root.probs.append(Pair(NEG_INF, cur))
        for t in range(envelope[0][1]):      # a bound past the end of read 2 is a slice panic in the reference
            cur = self.L.mul(cur, net2[t][0])
            root.probs.append(Pair(NEG_INF, cur))
        rows2 = lambda idx: net2[idx]
        data_of = lambda node: tree.get_data(node) if node != ROOT_NODE else root
        network_2_len = len(net2)
        last_upper_bound = 0
        for labelling_probs, bounds in zip(net1, envelope):
            next_beam = []
            lower_t, upper_t = max(bounds[0], 0), min(bounds[1], network_2_len)
            if lower_t >= upper_t or lower_t > last_upper_bound:
                raise SearchError(INVALID_ENVELOPE)
            if upper_t > last_upper_bound:
                beam = stable_sort_by_node(beam)            # parents before children
                for sp in beam:
                    info = tree.info(sp.node)
                    if info is not None:
                        label, parent = info
                        self.extend(tree.get_data(sp.node), rows2, data_of(parent), label,
                                    tree.label(parent) == label, lower_t, upper_t)
            last_upper_bound = upper_t
            for tip in beam:
                tip_label = tree.label(tip.node)
                if labelling_probs[0] > thr:
                    next_beam.append(Point2(tip.node, tip.state,
                                            Pair(NEG_INF, self.L.mul(self.pair_probability(tip.prob_1), labelling_probs[0])),
                                            tip.prob_2_max))
                for label in range(alphabet_size):
                    prob = labelling_probs[label + 1]
                    if prob < thr:
                        continue
                    if collapse_repeats and label == tip_label:
                        next_beam.append(Point2(tip.node, tip.state, Pair(self.L.mul(tip.prob_1.label, prob), NEG_INF),
                                                tip.prob_2_max))
                        new_node = tree.get_child(tip.node, label)
                        if new_node is None and tip.prob_1.gap > NEG_INF:
                            new_node = tree.add_node(tip.node, label,
                                                     self.build(rows2, data_of(tip.node), label, True, lower_t, upper_t))
                        if new_node is not None:
                            next_beam.append(Point2(new_node, tip.state, Pair(self.L.mul(tip.prob_1.gap, prob), NEG_INF),
                                                    tip.prob_2_max))
                    else:
                        new_node = tree.get_child(tip.node, label)
                        if new_node is None:
                            new_node = tree.add_node(tip.node, label,
                                                     self.build(rows2, data_of(tip.node), label, False, lower_t, upper_t))
                        next_beam.append(Point2(new_node, tip.state,
                                                Pair(self.L.mul(self.pair_probability(tip.prob_1), prob), NEG_INF),
                                                tip.prob_2_max))
            beam = self.merge_and_prune(next_beam, tree, beam_size)
        tokens = [alphabet[label + 1] for label, _ in tree.iter_from(beam[0].node)]
        tokens.reverse()
        return "".join(tokens)

    # :652-834
    def crf_beam_search(self, net1_real, init_state_1, net2_real, init_state_2, alphabet, envelope, beam_size, thr_real):
        net1 = [[[Log.new(v) for v in st] for st in row] for row in net1_real]
        net2 = [[[Log.new(v) for v in st] for st in row] for row in net2_real]
        thr = Log.new(f32(thr_real))
        n_state = len(net1[0])
        n_base = len(net1[0][0]) - 1

        def argmax(v):
            best = 0
            for i, e in enumerate(v):
                if e > v[best]:
                    best = i
            return best

        tree = SuffixTree(n_base)
        beam = [Point2(ROOT_NODE, argmax(init_state_1), Pair(NEG_INF, 0.0), 0.0)]
        # crf_root_probs (:411-441)
        root = Secondary(-1)
        root.max_prob = 0.0
        cur = 0.0
        root.probs.append(Pair(NEG_INF, cur))
        state = argmax(init_state_2)
        for t in range(envelope[0][1]):
            cur = self.L.mul(cur, net2[t][state][0])
            root.probs.append(Pair(NEG_INF, cur))
            state = (state * n_base) % n_state
        data_of = lambda node: tree.get_data(node) if node != ROOT_NODE else root
        network_2_len = len(net2)
        last_upper_bound = 0
        for probs, bounds in zip(net1, envelope):
            next_beam = []
            lower_t, upper_t = max(bounds[0], 0), min(bounds[1], network_2_len)
            if lower_t >= upper_t or lower_t > last_upper_bound:
                raise SearchError(INVALID_ENVELOPE)
            if upper_t > last_upper_bound:
                beam = stable_sort_by_node(beam)
                for sp in beam:
                    info = tree.info(sp.node)
                    if info is not None:
                        label, parent = info
                        tstate = sp.state                    # the ENTRY's state (:728)
                        self.extend(tree.get_data(sp.node), lambda idx: net2[idx][tstate], data_of(parent), label,
                                    False, lower_t, upper_t)
            last_upper_bound = upper_t
            for tip in beam:
                labelling_probs = probs[tip.state]           # an out-of-range state is an index panic
                if labelling_probs[0] > thr:
                    next_beam.append(Point2(tip.node, tip.state,
                                            Pair(NEG_INF, self.L.mul(self.pair_probability(tip.prob_1), labelling_probs[0])),
                                            tip.prob_2_max))
                for label in range(n_base):
                    prob = labelling_probs[label + 1]
                    if prob < thr:
                        continue
                    new_node = tree.get_child(tip.node, label)
                    if new_node is None:
                        tstate = tip.state                   # the TIP's state (:766)
                        new_node = tree.add_node(tip.node, label,
                                                 self.build(lambda idx: net2[idx][tstate], data_of(tip.node), label,
                                                            False, lower_t, upper_t))
                    next_beam.append(Point2(new_node, (tip.state * n_base) % n_state + label,
                                            Pair(self.L.mul(self.pair_probability(tip.prob_1), prob), NEG_INF),
                                            tip.prob_2_max))
            beam = self.merge_and_prune(next_beam, tree, beam_size)
        sequence = ""
        for label, _ in tree.iter_from(beam[0].node):
            sequence += alphabet[label + 1]
        return sequence[::-1]
