"""Parts of bench.py that are not the timed region: the self-launcher of the N > 1 run, repeat statistics."""
