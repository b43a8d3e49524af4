"""TEST INFRASTRUCTURE ONLY.

CPU restatement (numpy + plain C) of the reference's element-integration /
assembly path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import or execute anything under this
directory -- and there only as the checker, never as the thing that is measured
or shipped.  ``mimi_amd`` (the product) never imports it.
"""
