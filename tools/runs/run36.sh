set -e
time python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('ok')" 2>&1 | tail -4
python - <<'PY'
import json,sys
sys.path.insert(0,'.')
import weclip_vit_comer_amd.build as b
print(json.load(open('profiles/r04_traffic.json'))['__meta__']['source_hash'], b.source_hash())
PY
