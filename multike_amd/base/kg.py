"""Import-path compatibility with the reference (code/base/kg.py): the containers live in base/kgs.py."""
from .kgs import KG, parse_triples  # noqa: F401
