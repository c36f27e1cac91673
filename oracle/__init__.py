"""CPU oracle for the bi-date Siamese U-Net hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``fabric_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg do, and there only as the checker / reported baseline.

Parity status: PINNED by golden vectors generated in the build container by
importing the reference itself (``oracle/make_golden.py`` ->
``tests/golden/*.npz``); the reference ships no tests or golden vectors of its
own, so those fixtures are the only pin that exists.
"""
