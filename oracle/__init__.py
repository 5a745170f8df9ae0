"""CPU oracle for the Pix2Pix side2side training step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the timed CPU baseline.

PARITY UNPINNED: TensorFlow 2.9.1 / keras 2.9.0 / tensorflow-addons 0.17.1 (the
libraries that hold the reference's arithmetic, requirements.txt:46,99,100) are
not installed in the build container and the reference ships no tests, golden
vectors or fixtures for this path.  The oracle therefore
restates the published algorithms of those libraries at the reference's own
call sites and is pinned only by (a) structural known answers the reference
prints (parameter counts experiments.ipynb:198-199, layer shapes
networks.py:45-47,58-75), (b) analytic known answers, (c) an independent
pure-numpy index-level restatement (``np_restatement``) of every op.
"""
