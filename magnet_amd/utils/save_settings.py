"""Record of a run's command-line arguments beside its logs.

Drop-in for the helper the reference drivers call after training a split
(reference src/utils/save_settings.py:3-8; called at src/sparse_Magnet.py:199 and
src/Edge_sparseMagnet.py:219): `<log dir>/settings.csv`, one `name,value` row per
argument, in the order argparse defined them.
"""
import csv
import os


def write_log(args, path):
    """args: mapping name -> value (the drivers pass vars(namespace)); path: the run's log directory."""
    rows = [(str(name), value) for name, value in dict(args).items()]
    target = os.path.join(path, 'settings.csv')
    with open(target, 'w', newline='') as handle:
        csv.writer(handle).writerows(rows)
    return target
