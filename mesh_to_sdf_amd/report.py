"""Parity report between two signed-distance arrays of the same grid (pure numpy; no oracle import).

Used by bench.py (`reference_parity`) and the `-m gpu` tests to state what a user switching from the
reference sees: this library returns the EXACT minimum over all triangles, the reference's grid path a
heap-ordered label propagation (generate/grid.rs:495-558) that ends >= the exact minimum on part of the
cells (SURVEY.md header fact 2, §8c "Exact-vs-propagation policy").
"""
import numpy as np

TOL = 1e-5   # BASELINE.json north_star: absolute tolerance on the f32 distances


def reference_parity(ours, ref, normal_sign: bool = False) -> dict:
    """`ours`: this library's grid, `ref`: the reference semantics' grid (same layout).

    pct_within_1e-5      cells with | |ours| - |ref| | <= 1e-5
    pct_bit_identical    cells whose f32 bit patterns agree
    max_dev              max | |ours| - |ref| |
    p999_dev             99.9th percentile of the same
    sign_mismatches      cells whose sign bits differ
    ours_le_ref_everywhere  |ours| <= |ref| (+ the compare_distances tie window 1e-6 + 2 ulp in Normal mode,
                            lib.rs:242-259: the fold may keep a positive distance that is approx_eq to a smaller one)
    """
    a = np.ascontiguousarray(ours, np.float32).reshape(-1)
    b = np.ascontiguousarray(ref, np.float32).reshape(-1)
    assert a.shape == b.shape
    ma, mb = np.abs(a), np.abs(b)
    dev = np.abs(ma.astype(np.float64) - mb.astype(np.float64))
    slack = (1e-6 + 4 * np.spacing(mb).astype(np.float64)) if normal_sign else 0.0
    return {
        "cells": int(a.size),
        "pct_within_1e-5": round(100.0 * float(np.mean(dev <= TOL)), 4),
        "pct_bit_identical": round(100.0 * float(np.mean(a.view(np.uint32) == b.view(np.uint32))), 4),
        "max_dev": float(dev.max()) if a.size else 0.0,
        "p999_dev": float(np.quantile(dev, 0.999)) if a.size else 0.0,
        "sign_mismatches": int(np.count_nonzero(np.signbit(a) != np.signbit(b))),
        "ours_le_ref_everywhere": bool(np.all(ma.astype(np.float64) <= mb.astype(np.float64) + slack)),
    }
