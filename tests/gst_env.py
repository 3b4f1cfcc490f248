"""Environment for running the container's GStreamer 1.14 (/opt/conda) with the vfhip plugin (tests only)."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLUGIN_DIR = os.path.join(ROOT, "gstreamer-metal_amd", "gst")
PLUGIN = os.path.join(PLUGIN_DIR, "libgstvfhip.so")
GST_LAUNCH = "/opt/conda/bin/gst-launch-1.0"
GST_INSPECT = "/opt/conda/bin/gst-inspect-1.0"
_REG = os.path.join(tempfile.gettempdir(), f"gst-registry-vfhip-{os.getuid()}.bin")


def available():
    return os.path.exists(GST_LAUNCH) and os.path.exists(PLUGIN)


def env():
    e = dict(os.environ)
    e.update(PATH="/opt/conda/bin:" + e.get("PATH", ""), GST_PLUGIN_SYSTEM_PATH="/opt/conda/lib/gstreamer-1.0",
             GST_PLUGIN_SCANNER="/opt/conda/libexec/gstreamer-1.0/gst-plugin-scanner", GST_REGISTRY=_REG,
             GST_PLUGIN_PATH=PLUGIN_DIR, LD_LIBRARY_PATH="/opt/conda/lib",
             # conda's libstdc++ is older than what libamdhip64 needs: take the system one
             LD_PRELOAD="/usr/lib/x86_64-linux-gnu/libstdc++.so.6")
    return e


def launch(pipeline, timeout=120, verbose=False, debug=None):
    """verbose: gst-launch -v (prints every pad's negotiated caps) instead of -q; debug: a GST_DEBUG spec (log on stderr)"""
    e = env()
    if debug:
        e.update(GST_DEBUG=debug, GST_DEBUG_NO_COLOR="1")
    return subprocess.run(f"{GST_LAUNCH} {'-v' if verbose else '-q'} {pipeline}", shell=True, env=e, capture_output=True, text=True, timeout=timeout)


def inspect(what):
    return subprocess.run([GST_INSPECT, what], env=env(), capture_output=True, text=True, timeout=60)
