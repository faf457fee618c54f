for i in 1 2 3 4 5; do python -m pytest tests -q -m gpu -rf -p no:cacheprovider 2>&1 | grep -E "^FAILED|Max absolute|Max relative|Mismatched|passed|failed" ; done
