"""Darknet-style .cfg reader — same contract as the reference's CVC-YOLOv3/utils/parse_config.py:1-18:
a list of dicts of *strings*, one per ``[section]``, key ``type`` = section name, ``batch_normalize`` defaulted to 0
on convolutional sections, lines starting with '#' and empty lines skipped, whitespace around keys/values stripped."""


def parse_model_config(path):
    with open(path, "r") as fh:
        rows = fh.read().split("\n")
    sections = []
    for row in rows:
        if row == "" or row[0] == "#":
            continue
        row = row.strip()
        if row.startswith("["):
            kind = row[1:-1].rstrip()
            sections.append({"type": kind})
            if kind == "convolutional":
                sections[-1]["batch_normalize"] = 0
            continue
        key, value = row.split("=")
        sections[-1][key.rstrip()] = value.strip()
    return sections
