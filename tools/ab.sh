# same-box A/B of the default library against variants: window(2000 iterations) time, alternating.  usage: bash tools/ab.sh base [base2 ...]
for rep in 1 2; do
  echo -n "default: "; python tools/window.py 2000 2 2>&1 | grep window
  for v in "$@"; do echo -n "$v: "; LPBOX_LIB_VARIANT=$v python tools/window.py 2000 2 2>&1 | grep window; done
done
