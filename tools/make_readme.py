"""Regenerates tools/README.md: one table row per script, taken from its docstring or leading comment.   python tools/make_readme.py"""
import ast
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEAD = '''# tools/ — measurement and diagnosis scripts (none of them is on the product path)

Interleaved A/B timings of kernel structures and step-level switches, per-layer tables, rocprofv3 summaries, in-kernel cycle stamps (diagnostic builds),
timing ablations (variant builds: `python continual-learning_amd/build.py --variant NAME -DFLAG`, selected with `CLAMD_LIB=build/NAME/libclamd.so`),
parity diagnostics.  They import `continual_learning_amd` only and run on one MI355X; the two diagnostics that compare against the stock torch counterpart
(`oracle/`) live under `tests/diag/` (`grad_accuracy.py`, `fold_two_step.py`): only `tests/`, `smoke()` and `bench.py`'s CPU baseline touch the oracle.

| script | what it does |
|---|---|
'''


def describe(path):
    src = open(path).read()
    if path.endswith('.py'):
        try:
            doc = ast.get_docstring(ast.parse(src)) or ''
        except SyntaxError:
            doc = ''
    else:
        lines = []
        for l in src.splitlines():
            if l.startswith('#!'):
                continue
            if l.startswith(('#', '//')):
                lines.append(l.lstrip('#/ ').rstrip())
            elif lines:
                break
        doc = ' '.join(lines)
    doc = re.sub(r'\s+', ' ', doc.split('\n\n')[0]).strip().replace('|', '/')
    return doc if len(doc) <= 300 else doc[:297] + '...'


rows = [f'| `{f}` | {describe(os.path.join(HERE, f))} |' for f in sorted(os.listdir(HERE))
        if f.endswith(('.py', '.sh')) and f != 'make_readme.py']
ub = [f'| `ubench/{f}` | {describe(os.path.join(HERE, "ubench", f))} |' for f in sorted(os.listdir(os.path.join(HERE, 'ubench'))) if f.endswith('.hip')]
open(os.path.join(HERE, 'README.md'), 'w').write(HEAD + '\n'.join(rows) + '\n\n`ubench/`: stand-alone HIP micro-benchmarks and reproducers behind the design '
                                                 'decisions in DESIGN.md §4 (build with `hipcc --offload-arch=gfx950 -O3`).\n\n| file | what it shows |\n|---|---|\n' + '\n'.join(ub) + '\n')
