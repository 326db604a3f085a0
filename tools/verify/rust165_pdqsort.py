"""A REAL rustc-compiled pattern-defeating quicksort, found in this image, as a second opinion on the restatement.

    python tools/verify/rust165_pdqsort.py [tools/verify/pdq178_vectors.json]

No Rust toolchain exists here, but a Rust standard library does -- compiled: libcst's native module (libcst 0.4.9,
`rustc 1.65.0`, commit 897e37553bba in its panic paths) carries two monomorphisations of `core::slice::sort::recurse`
with their local symbols -- the body of `sort_unstable_by_key`:

    recurse<T, F>(v: &mut [T], is_less: &mut F, pred: Option<&T>, limit: u32)      (rdi, rsi | rdx | rcx | r8d)

one over 24-byte elements in ascending order of their first u64, one over 16-byte elements in DESCENDING order of their
second u64 (read off
the disassembly: `cmp $0x15,%rsi` -- insertion sort up to 20 elements; `cmp $0x31` -- the ninther from 50; the compares
themselves).  The comparator is inlined and the rest of an element is payload, so calling the routine on (key, node)
records whose keys order like the reference's comparator (src/search.rs:262-269: greater probability first, equal
probabilities equal) runs Rust's own quicksort on the reference's question -- the order of EQUAL keys included.

What it can and cannot say.  std's unstable sort was this pdqsort from 1.20 to 1.80 (1.81 replaced it).  Between 1.65
and the reference's 1.78 the authors know of TWO changes that can move elements, both of early 2023:
  * `break_patterns` drew its `usize` numbers as two 32-bit xorshift values (13, 17, 5) and draws them from one
    usize-wide xorshift (64-bit: 13, 7, 17; `seed = len`) since;
  * `partial_insertion_sort`, having swapped the out-of-order pair, called shift_tail(&mut v[..i]) and
    shift_head(&mut v[i..]); since the refactor of the insertion sorts it calls insertion_sort_shift_left(&mut v[..i],
    i - 1) and insertion_sort_shift_right(&mut v[..i], 1) -- both on v[..i].
The oracle restates the LATER forms (Rust 1.78 as recalled; so do the kernels) and keeps the earlier ones behind
`fcdo_set_pdq_std_form(bits)` (the kernels: FCD_PDQ178_STD_FORM, include/fcd.h).  This program sorts every committed vector with the compiled 1.65 routine and compares:
with both earlier forms selected the restatement must equal it on every list, element for element -- which pins pivot
choice, partition_in_blocks, partition_equal, the recursion with its limit and flags, heapsort, the reversal, the
insertion sorts and the earlier forms themselves to a rustc-compiled std -- and under the default forms every
difference must be on a list that reaches one of the two changed routines.  What remains for pdq178_check.rs (rustc 1.78
proper) is whether 1.78 carries the later forms; the vectors say per list which form gives which permutation.

The oracle can also run its SEARCHES on the compiled routine (fcdo_set_external_recurse: every list above 20 candidates
is sorted by Rust's own code): `--searches` decodes the BASELINE configurations that way and compares with the oracle on
its restatement under both forms.

Test infrastructure only (tests/test_rust165_pdqsort.py runs it when the module is there); nothing in the product or in
the oracle depends on libcst."""
import ctypes
import importlib.util
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
RECURSE = "_ZN4core5slice4sort7recurse"


def native_path():
    spec = importlib.util.find_spec("libcst")
    if spec is None or not spec.submodule_search_locations:
        return None
    d = list(spec.submodule_search_locations)[0]
    for f in sorted(os.listdir(d)):
        if f.startswith("native") and f.endswith(".so"):
            return os.path.join(d, f)
    return None


def rustc_commit(path):
    data = open(path, "rb").read()
    i = data.find(b"/rustc/")
    return data[i + 7:i + 47].decode("ascii", "replace") if i >= 0 else ""


def recurse_symbols(path):
    try:
        out = subprocess.run(["nm", path], capture_output=True, text=True, timeout=60).stdout
    except (OSError, subprocess.SubprocessError):
        return []
    syms = []
    for line in out.splitlines():
        parts = line.split()
        if len(parts) == 3 and parts[2].startswith(RECURSE):
            syms.append((parts[2], int(parts[0], 16)))
    return sorted(syms)


def load_base(path):
    real = os.path.realpath(path)
    with open("/proc/self/maps") as f:
        for line in f:
            p = line.split()
            if len(p) >= 6 and os.path.realpath(p[5]) == real and int(p[2], 16) == 0:
                return int(p[0].split("-")[0], 16)
    return None


def other_modules():
    """other Rust-built extension modules of this image known to carry a pre-2023 core::slice::sort::recurse with its
    symbols (a second toolchain version for the same question): cryptography's binding in the image's conda environment
    (rustc commit a178d0322ce2, 2021: its comparator closure is zero-sized and not passed at all)"""
    import glob
    return sorted(glob.glob("/opt/conda/lib/python3*/site-packages/cryptography/hazmat/bindings/_rust.abi3.so"))


class Rust165Sort:
    """sort(keys: uint64[n], payload: int64[n]) -> payload in the order the compiled routine leaves it (ascending keys)"""

    # (element bytes, key offset, payload offset, the comparator is `a.key > b.key`: the instance sorts DESCENDING)
    LAYOUTS = ((24, 0, 8, False), (16, 8, 0, True))

    def __init__(self, path=None):
        self.path = path or native_path()
        self.fn = None
        self.layout = None
        self.symbol = None
        if not self.path:
            return
        syms = recurse_symbols(self.path)
        if not syms:
            return
        self._lib = ctypes.CDLL(self.path)  # (kept: the mapping must stay)
        base = load_base(self.path)
        if base is None:
            return
        # two shapes of the call: (v.ptr, v.len, &mut is_less, pred, limit), or without the closure when it is zero-sized
        proto = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32)
        proto_nc = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_uint32)
        self.candidates = [(name, proto(base + off)) for name, off in syms]
        self.candidates_nc = [(name, proto_nc(base + off)) for name, off in syms]
        self.no_closure = False
        self.addresses = {name: base + off for name, off in syms}

    def call(self, fn, layout, keys, payload):
        size, koff, poff, descending = layout
        n = len(keys)
        keys = np.ascontiguousarray(keys, np.uint64)
        if descending:  # the same comparison outcomes from the complemented keys
            keys = ~keys
        buf = np.zeros(n * size + 64, np.uint8)
        rec = buf[:n * size].reshape(n, size)
        rec[:, koff:koff + 8] = np.ascontiguousarray(keys, np.uint64).view(np.uint8).reshape(n, 8)
        rec[:, poff:poff + 8] = np.ascontiguousarray(payload, np.int64).view(np.uint8).reshape(n, 8)
        dummy = ctypes.create_string_buffer(64)  # is_less: the comparator is inlined, its environment is never read
        limit = int(n).bit_length()  # usize::BITS - len.leading_zeros()
        if self.no_closure:
            fn(buf.ctypes.data, n, None, limit)
        else:
            fn(buf.ctypes.data, n, ctypes.addressof(dummy), None, limit)
        out_k = rec[:, koff:koff + 8].copy().view(np.uint64).reshape(n)
        if descending:
            out_k = ~out_k
        out_p = rec[:, poff:poff + 8].copy().view(np.int64).reshape(n)
        return out_k, out_p

    def probe(self, index, layout_index):
        """(in a child process: a wrong guess may crash) does candidate `index` sort records of this layout?"""
        name, fn = (self.candidates_nc if self.no_closure else self.candidates)[index]
        rng = np.random.default_rng(5)
        for n in (30, 77, 300):
            keys = rng.permutation(n).astype(np.uint64) * 3 + 1
            pay = np.arange(n, dtype=np.int64) + 1000
            k, p = self.call(fn, self.LAYOUTS[layout_index], keys, pay)
            if not (np.all(k[:-1] <= k[1:]) and np.array_equal(keys[p - 1000], k)):
                return False
        # a call of the wrong shape may still sort -- by heapsort, with the limit read as 0: it must be the quicksort
        keys = np.array([3, 3, 0, 3, 2, 1, 2, 0, 0, 1, 2, 2, 2, 0, 0, 1, 3, 0, 3, 3, 2], np.uint64)
        want = [8, 17, 2, 14, 13, 7, 15, 5, 9, 6, 10, 11, 12, 4, 20, 1, 0, 16, 3, 18, 19]
        _, p = self.call(fn, self.LAYOUTS[layout_index], keys, np.arange(21, dtype=np.int64))
        if self.LAYOUTS[layout_index][3]:
            return True  # (the descending instance orders equal keys its own way: checked against the restatement later)
        return p.tolist() == want

    def select(self, skip=0):
        """finds a (routine, layout) pair that behaves -- the (skip + 1)-th one; each attempt runs in a child process first"""
        if not getattr(self, "candidates", None):
            return False
        for ci in range(len(self.candidates)):
            found = False
            for li in range(len(self.LAYOUTS)):
                for nc in (0, 1):
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--probe", str(ci), str(li), str(nc), self.path],
                                       capture_output=True, text=True, timeout=120)
                    if r.returncode == 0 and r.stdout.strip().endswith("ok"):
                        found = True
                        if skip > 0:
                            skip -= 1
                            break  # (the next routine: a second layout of this one is not a second instance)
                        self.no_closure = bool(nc)
                        self.symbol, self.fn = (self.candidates_nc if nc else self.candidates)[ci]
                        self.layout = self.LAYOUTS[li]
                        return True
                if found:
                    break
        return False

    def sort(self, keys, payload):
        return self.call(self.fn, self.layout, keys, payload)[1]

    def address_of_ascending_24(self):
        """the address of the instance over 24-byte records in ascending order of their first word (what the oracle's
        fcdo_set_external_recurse expects), or None"""
        for skip in (0, 1, 2):
            t = Rust165Sort(self.path)
            if not t.select(skip):
                return None
            if t.layout == self.LAYOUTS[0] and not t.no_closure:
                self._keep = t  # (its mapping of the module stays)
                return t.addresses[t.symbol]
        return None


def keys_of(p):
    """u64 keys that order like src/search.rs:262-269: a is `less` than b when its probability is GREATER; equal
    probabilities (-0.0 and +0.0 included) are equal"""
    p = np.ascontiguousarray(p, np.float32) + np.float32(0.0)
    u = p.view(np.uint32).astype(np.uint64)
    u = np.where(u & np.uint64(0x80000000), u ^ np.uint64(0xFFFFFFFF), u | np.uint64(0x80000000))  # ascending with p
    return np.uint64(0xFFFFFFFF) - u


def main(argv):
    if len(argv) >= 4 and argv[1] == "--probe":
        s = Rust165Sort(argv[5] if len(argv) > 5 else None)
        s.no_closure = len(argv) > 4 and argv[4] == "1"
        ok = bool(getattr(s, "candidates", None)) and s.probe(int(argv[2]), int(argv[3]))
        print("ok" if ok else "no")
        return 0 if ok else 1
    if len(argv) >= 2 and argv[1] == "--searches":
        return searches(int(argv[2]) if len(argv) > 2 else 1024)
    path = argv[1] if len(argv) > 1 else os.path.join(ROOT, "tools", "verify", "pdq178_vectors.json")
    s = Rust165Sort()
    if not s.path:
        print("libcst's native module is not installed here: nothing to compare with")
        return 2
    if not s.select():
        print("%s: no core::slice::sort::recurse that takes (key, payload) records found" % s.path)
        return 2
    print("%s\n  rustc commit %s" % (s.path, rustc_commit(s.path)))
    ok, skip = True, 0
    while True:  # every compiled instance of the routine that takes (key, payload) records
        print("%s, %d-byte elements, %s order of the key" % (s.symbol, s.layout[0], "descending" if s.layout[3] else "ascending"))
        rep = compare(s, path)
        for line in rep["lines"]:
            print("  " + line)
        ok = ok and rep["ok"]
        skip += 1
        s = Rust165Sort()
        if not s.select(skip):
            break
    for extra in other_modules():  # a second toolchain version, if the image has one
        s = Rust165Sort(extra)
        if s.select():
            print("%s\n  rustc commit %s\n%s, %d-byte elements, %s order of the key" % (extra, rustc_commit(extra), s.symbol, s.layout[0],
                  "descending" if s.layout[3] else "ascending"))
            rep = compare(s, path)
            for line in rep["lines"]:
                print("  " + line)
            ok = ok and rep["ok"]
    return 0 if ok else 1


def searches(n_config3):
    """BASELINE config 2 (all 4096 reads), its rows at beam 12, and n_config3 reads of config 3 (beam 32): the oracle with
    the compiled rustc-1.65 quicksort in place of its restatement, against the oracle under the earlier / later std forms"""
    import time
    sys.path.insert(0, ROOT)
    import bench
    from oracle import oracle
    s = Rust165Sort()
    if not s.path or not s.select():
        print("no compiled core::slice::sort::recurse here")
        return 2
    addr = s.address_of_ascending_24()
    if addr is None:
        print("no instance over 24-byte ascending records")
        return 2

    def run(x, beam):
        n, T = x.shape[0], x.shape[1]
        out = oracle.batch_outputs(n, T)
        lab, path, lens, st = oracle.beam_search_batch(x, beam, 0.1, True, os.cpu_count() or 1, out=out)
        return [(int(st[i]), lab[i, :lens[i]].tobytes(), path[i, :lens[i]].tobytes()) for i in range(n)]

    bad = 0
    for name, x, beam in (("BASELINE config 2 (4096 reads, beam 5)", bench.make_batch(1, 4096), 5),
                          ("config 2's rows at beam 12 (1024 reads)", bench.make_batch(1, 1024), 12),
                          ("BASELINE config 3 (%d reads, beam 32)" % n_config3, bench.make_batch(1, n_config3), 32)):
        t0 = time.time()
        with oracle.unstable_sort("pdqsort"):
            with oracle.external_recurse(addr):
                real = run(x, beam)
            with oracle.pdq_std_form(3):
                earlier = run(x, beam)
            later = run(x, beam)
        d_e = sum(a != b for a, b in zip(real, earlier))
        d_l = sum(a != b for a, b in zip(real, later))
        bad += d_e
        print("%s: the oracle ON THE COMPILED rustc-1.65 quicksort vs the oracle on its restatement -- earlier std forms: %d reads "
              "differ; later forms (the default, Rust 1.78 as recalled): %d reads differ  (%.0f s)" % (name, d_e, d_l, time.time() - t0),
              flush=True)
    return 0 if bad == 0 else 1


FORMS = ((3, "both routines as until 2022 (= what rustc 1.65 compiled)"), (1, "the generator as until 2022 only"),
         (2, "partial_insertion_sort as until 2022 only"), (0, "both as since 2023: Rust 1.78 as recalled, the default"))


def compare(s, path=None):
    """-> {"ok": the pinning statements hold, "lines": the report, "differ": {form: [case indices]}}"""
    sys.path.insert(0, ROOT)
    from oracle import oracle
    doc = json.load(open(path or os.path.join(ROOT, "tools", "verify", "pdq178_vectors.json")))
    cases = doc["cases"]
    real = []
    for c in cases:
        p = np.array(c["bits"], np.uint32).view(np.float32)
        real.append(s.sort(keys_of(p), np.arange(len(p), dtype=np.int64)))
    differ, reach = {}, {}
    for form, _ in FORMS:
        differ[form], reach[form] = [], []
        for i, c in enumerate(cases):
            p = np.array(c["bits"], np.uint32).view(np.float32)
            with oracle.unstable_sort("pdqsort"), oracle.pdq_std_form(form):
                oracle.pdq_path_counts(reset=True)
                _, perm = oracle.pdqsort_desc(p, np.arange(len(p), dtype=np.int32))
                reach[form].append(oracle.pdq_path_counts())
            if not np.array_equal(real[i], perm):
                differ[form].append(i)
    lines = ["%d lists; %d reach break_patterns, %d shift in partial_insertion_sort (default forms)"
             % (len(cases), sum(1 for b, _ in reach[0] if b), sum(1 for _, q in reach[0] if q))]
    for form, what in FORMS:
        lines.append("restatement with %s: %d lists differ from the compiled routine" % (what, len(differ[form])))
    # every difference under the default forms is explained by a changed routine the list reaches
    unexplained = [i for i in differ[0] if not (reach[0][i][0] or reach[0][i][1])]
    only_gen = [i for i in differ[1] if not reach[1][i][1]]      # generator old, shifting new: a difference needs a shift
    only_shift = [i for i in differ[2] if not reach[2][i][0]]    # shifting old, generator new: ... needs break_patterns
    lines.append("differences under the default forms on lists that reach NEITHER changed routine: %d" % len(unexplained))
    ok = not differ[3] and not unexplained and not only_gen and not only_shift
    lines.append("PINNED to this compiled std except for the two routines std changed in 2023" if ok
                 else "NOT pinned: see the counts above")
    return {"ok": ok, "lines": lines, "differ": differ}


if __name__ == "__main__":
    sys.exit(main(sys.argv))
