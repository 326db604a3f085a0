// pdq178_check.rs -- does THIS repository's restatement of Rust 1.78's `sort_unstable_by` leave equal probabilities
// in the order the real one does?
//
//     rustc +1.78.0 -O tools/verify/pdq178_check.rs -o /tmp/pdq178_check
//     /tmp/pdq178_check tools/verify/pdq178_vectors.json
//
// (any way of getting the 1.78.0 toolchain will do: `rustup toolchain install 1.78.0`; the reference pins it in
// .github/workflows/test.yml:16 and build-wheels.sh:6.  A different toolchain answers a different question: the
// standard library's unstable sort was replaced in 1.81.)
//
// The vectors (tools/verify/make_pdq178_vectors.py) are lists of (probability, node) pairs in ascending node order --
// what fast-ctc-decode's prune hands to the sort after its stable sort by node (src/search.rs:245-260) -- each with
// the permutation the restatement (oracle/fcd_oracle.c DEFINE_PDQSORT; the GPU kernels' csrc/pdq178.h and
// csrc/pdq178_wave.h are tested element for element against it) produces.  This program sorts every list with the
// reference's own comparator (src/search.rs:262-269, the same closure at src/duplex.rs:620 and :807 and
// src/search.rs:122) and compares.  No such run could be made where the restatement was written (no rustc in
// that image): the claim "follows Rust 1.78" rests, for two routines, on recollection until somebody runs this (the
// rest is pinned against a compiled rustc-1.65 std, below).  Exit status 0 = every list agrees; 1 = mismatches (each
// one printed: case index, length, and which earlier form of std it matches instead, if any).
//
// What else the file knows.  A rustc-1.65 build of std exists, compiled, in the image the restatement was written in
// (libcst's native module); tools/verify/rust165_pdqsort.py calls its core::slice::sort::recurse on these lists.  With
// the EARLIER forms of the two routines std changed in 2023 -- break_patterns drawing two 32-bit xorshift numbers per
// usize, partial_insertion_sort calling shift_tail(&mut v[..i]) / shift_head(&mut v[i..]) -- the restatement equals that
// binary on every list; `perm` follows the LATER forms (one usize-wide xorshift; insertion_sort_shift_left /
// insertion_sort_shift_right on v[..i]), which is Rust 1.78 as recalled.  Where the earlier forms give another
// permutation the case carries it: "perm_g" (earlier generator), "perm_p" (earlier partial_insertion_sort), "perm_gp"
// (both).  A mismatch with `perm` is therefore also compared with those, and the summary says which form THIS
// toolchain's std agrees with -- a toolchain from before 2023 is expected to agree with perm_gp everywhere.  The
// product follows whichever is asked for: FCD_PDQ178_STD_FORM = 0 (perm, default), 1 (perm_g), 2 (perm_p), 3 (perm_gp)
// in the environment when the library is loaded (include/fcd.h).
//
// No dependencies: the file is parsed by a scanner that knows its shape
//     {"meta": {...}, "cases": [{"bits": [u32, ...], "perm": [int, ...] (, "perm_g": [...], "perm_p": [...], "perm_gp": [...])}, ...]}
use std::env;
use std::fs;
use std::process;

#[derive(Clone, Copy)]
struct SearchPoint {
    node: i32,
    label_prob: f32,
    gap_prob: f32,
}

impl SearchPoint {
    // src/search.rs:24-27
    fn probability(&self) -> f32 {
        self.label_prob + self.gap_prob
    }
}

// the integers of the array that follows the next occurrence of `"key":` at or after `from`
fn next_array(text: &[u8], from: usize, key: &str) -> Option<(Vec<u64>, usize)> {
    let pat = format!("\"{}\":", key);
    let pat = pat.as_bytes();
    let mut i = from;
    let start = loop {
        if i + pat.len() > text.len() {
            return None;
        }
        if &text[i..i + pat.len()] == pat {
            break i + pat.len();
        }
        i += 1;
    };
    let mut i = start;
    while i < text.len() && text[i] != b'[' {
        i += 1;
    }
    i += 1;
    let mut out = Vec::new();
    let mut cur: Option<u64> = None;
    while i < text.len() && text[i] != b']' {
        let c = text[i];
        if c.is_ascii_digit() {
            cur = Some(cur.unwrap_or(0) * 10 + (c - b'0') as u64);
        } else if let Some(v) = cur.take() {
            out.push(v);
        }
        i += 1;
    }
    if let Some(v) = cur.take() {
        out.push(v);
    }
    Some((out, i + 1))
}

// where the next occurrence of `pat` at or after `from` starts (text.len() if there is none)
fn find_from(text: &[u8], from: usize, pat: &[u8]) -> usize {
    let mut i = from;
    while i + pat.len() <= text.len() {
        if &text[i..i + pat.len()] == pat {
            return i;
        }
        i += 1;
    }
    text.len()
}

// the optional array `key` of the case that ends before `case_end`
fn optional_array(text: &[u8], from: usize, case_end: usize, key: &str) -> Option<Vec<u64>> {
    let pat = format!("\"{}\":", key);
    let at = find_from(text, from, pat.as_bytes());
    if at >= case_end {
        return None;
    }
    next_array(text, at, key).map(|(v, _)| v)
}

fn main() {
    let path = env::args().nth(1).unwrap_or_else(|| "tools/verify/pdq178_vectors.json".to_string());
    let text = fs::read(&path).unwrap_or_else(|e| {
        eprintln!("cannot read {}: {}", path, e);
        process::exit(2);
    });
    // skip the meta object: the cases start at "cases": [
    let cases_at = {
        let pat = b"\"cases\": [";
        let mut at = None;
        let mut i = 0;
        while i + pat.len() <= text.len() {
            if &text[i..i + pat.len()] == pat {
                at = Some(i + pat.len());
            }
            i += 1;
        }
        at.unwrap_or_else(|| {
            eprintln!("{}: no \"cases\": [ found", path);
            process::exit(2);
        })
    };
    let mut pos = cases_at;
    let mut n_cases = 0usize;
    let mut n_bad = 0usize;
    let mut n_unstable = 0usize;
    // lists that carry an alternative, and on how many of those this toolchain produced it
    let alt_keys = ["perm_g", "perm_p", "perm_gp"];
    let mut alt_present = [0usize; 3];
    let mut alt_agrees = [0usize; 3];
    let mut n_bad_unexplained = 0usize;
    while let Some((bits, after_bits)) = next_array(&text, pos, "bits") {
        let (perm, after_perm) = next_array(&text, after_bits, "perm").unwrap_or_else(|| {
            eprintln!("case {}: \"bits\" without \"perm\"", n_cases);
            process::exit(2);
        });
        pos = after_perm;
        let case_end = find_from(&text, after_perm, b"\"bits\":");
        let alts: Vec<Option<Vec<u64>>> = alt_keys.iter().map(|k| optional_array(&text, after_perm, case_end, k)).collect();
        if bits.len() != perm.len() {
            eprintln!("case {}: {} probabilities, {} permutation entries", n_cases, bits.len(), perm.len());
            process::exit(2);
        }
        // the list as the prune holds it: ascending node order (src/search.rs:245), probability = label + gap
        let mut beam: Vec<SearchPoint> = bits
            .iter()
            .enumerate()
            .map(|(i, &b)| SearchPoint { node: i as i32, label_prob: f32::from_bits(b as u32), gap_prob: 0.0 })
            .collect();
        let mut stable = beam.clone();
        // src/search.rs:262-269, verbatim but for the name of the flag
        let mut has_nans = false;
        beam.sort_unstable_by(|a, b| {
            (b.probability())
                .partial_cmp(&(a.probability()))
                .unwrap_or_else(|| {
                    has_nans = true;
                    std::cmp::Ordering::Equal // don't really care
                })
        });
        if has_nans {
            eprintln!("case {}: the vectors hold no NaNs, yet the comparator met one", n_cases);
            process::exit(2);
        }
        stable.sort_by(|a, b| b.probability().partial_cmp(&a.probability()).unwrap());
        if beam.iter().zip(stable.iter()).any(|(x, y)| x.node != y.node) {
            n_unstable += 1;
        }
        let got: Vec<u64> = beam.iter().map(|x| x.node as u64).collect();
        for a in 0..3 {
            if let Some(alt) = &alts[a] {
                alt_present[a] += 1;
                if *alt == got {
                    alt_agrees[a] += 1;
                }
            }
        }
        if let Some(j) = (0..beam.len()).find(|&j| got[j] != perm[j]) {
            n_bad += 1;
            let which = (0..3).find(|&a| alts[a].as_ref().map_or(false, |alt| *alt == got));
            match which {
                Some(a) => println!(
                    "MISMATCH case {} (length {}): rustc's sort_unstable_by produced \"{}\" (an EARLIER form of a routine std changed in 2023), not \"perm\"",
                    n_cases, beam.len(), alt_keys[a]
                ),
                None => {
                    n_bad_unexplained += 1;
                    println!(
                        "MISMATCH case {} (length {}): position {} holds node {} after rustc's sort_unstable_by, the restatement says {} (no listed alternative matches either)",
                        n_cases, beam.len(), j, beam[j].node, perm[j]
                    );
                }
            }
        }
        n_cases += 1;
    }
    println!(
        "{} lists, {} of them ordered differently from a stable sort by rustc's sort_unstable_by, {} mismatches with the restatement",
        n_cases, n_unstable, n_bad
    );
    for a in 0..3 {
        println!(
            "  \"{}\" (earlier {}): listed for {} lists, produced by this toolchain on {} of them",
            alt_keys[a],
            ["break_patterns generator", "partial_insertion_sort", "generator and partial_insertion_sort"][a],
            alt_present[a],
            alt_agrees[a]
        );
    }
    if n_bad > 0 {
        println!(
            "{} of the {} mismatches match no listed alternative{}",
            n_bad_unexplained,
            n_bad,
            if n_bad_unexplained == 0 { ": this std carries an earlier form of a routine the restatement has in its 2023 form (oracle/fcd_oracle.c fcdo_set_pdq_std_form; csrc/pdq178.h break_patterns / partial_insertion_sort are the two spots)" } else { "" }
        );
    }
    if n_cases == 0 {
        eprintln!("no cases found in {}", path);
        process::exit(2);
    }
    process::exit(if n_bad == 0 { 0 } else { 1 });
}
