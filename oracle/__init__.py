"""TEST INFRASTRUCTURE: CPU oracle (restatement of the reference pipeline). Importable only from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg — never from the product package."""
