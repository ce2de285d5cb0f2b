"""CPU oracle — test infrastructure only. Importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never from the product package."""
