for v in "$@"; do echo -n "$v: "; LPBOX_LIB_VARIANT=ko_$v python tools/window.py 1000 2 2>&1 | grep window; done
