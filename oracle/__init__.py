"""CPU oracle for the CurveCloudNet curve-aggregation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported, linked or executed by the
product package ``curvecloudnet_amd``; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may use it, and there only as the checker / reported
baseline.  See ``oracle/torch_ref.py`` for the restatement and ``oracle/gen_golden.py`` for how
it is pinned against the reference's own code (golden vectors in ``tests/golden``).
"""
