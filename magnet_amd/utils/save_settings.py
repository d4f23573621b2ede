"""settings.csv beside the logs of a run (reference src/utils/save_settings.py:3-8): one
`parameter,value` row per command-line argument."""
import csv


def write_log(args, path):
    with open(path + '/settings.csv', 'w', newline='') as f:
        w = csv.writer(f)
        for name in args:
            w.writerow([name, args[name]])
