#!/usr/bin/env python3
"""Golden vectors for the flat-array grounding (tests/golden/grounding.json.gz), from the reference's RelationalGraph.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference, read-only).  The two relational templates
are the reference's own generators (Demo/Data/RGM/Generator.py, Demo/Data/HMLN/GeneratorPaperPopularity.py), imported
and grounded with the reference's classes; only *data* is written: for every ground factor the index of its parametric
factor and the atom keys of its scope, plus the keys of all ground rvs.  usage: python oracle/capture_grounding.py
"""
import collections
import collections.abc
import contextlib
import gzip
import io
import json
import os
import sys
import time

import numpy as np

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
np.Inf = np.inf
collections.MutableSet = collections.abc.MutableSet
time.clock = time.perf_counter
sys.dont_write_bytecode = True
sys.path.insert(0, REF)


def dump(rel_g):
    g, rvs_dict = rel_g.ground_graph()
    key_of = {id(rv): list(k) for k, rv in rvs_dict.items()}
    pf_of = {id(pf.potential): i for i, pf in enumerate(rel_g.param_factors)}
    factors = sorted([pf_of[id(f.potential)], [key_of[id(rv)] for rv in f.nb]] for f in g.factors)
    return {'rvs': sorted(key_of.values()), 'factors': factors}


def main():
    out = {}
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            import importlib.util

            def load(name, path):
                spec = importlib.util.spec_from_file_location(name, path)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                return mod
            rgm = load('ref_rgm_generator', os.path.join(REF, 'Demo/Data/RGM/Generator.py'))
            out['rgm_100x10'] = dump(rgm.generate_rel_graph())
            hmln = load('ref_pp_generator', os.path.join(REF, 'Demo/Data/HMLN/GeneratorPaperPopularity.py'))
            out['paper_popularity_300x10'] = dump(hmln.generate_rel_graph())
    finally:
        os.chdir(cwd)
    path = os.path.join(ROOT, 'tests', 'golden', 'grounding.json.gz')
    with gzip.open(path, 'wt') as fh:
        json.dump(out, fh, separators=(',', ':'))
    print('wrote', path, os.path.getsize(path), 'bytes', {k: (len(v['rvs']), len(v['factors'])) for k, v in out.items()})


if __name__ == '__main__':
    main()
