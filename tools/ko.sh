for v in "$@"; do echo -n "$v: "; LPBOX_LIB_VARIANT=$v python tools/window.py 2000 2 2>&1 | grep window; done
