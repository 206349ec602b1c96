"""Static checks of integration/*.{h,cpp} against the reference's headers.

The binding files cannot be compiled in this image (they need jsoncpp and the generated load-parameters.h, and no stand-ins are
written); `tools/check_integration.sh <reference-tree>` is the compile + link + run check for a configured tree.  What CAN be kept
green here: every reference symbol the files use still exists, with that spelling, in the reference header it comes from -- a
renamed member or method in the reference fails this file -- and the mechanical edits of `enodeb-cases.inc` still apply to the
reference's ENodeB.{h,cpp} and scenario file.  Skipped where /root/reference is absent (the GPU box)."""
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")
SRC = REF / "src"
INTEG = ROOT / "integration"

pytestmark = pytest.mark.skipif(not SRC.is_dir(), reason="the reference tree is not on this machine")

SCHED = "protocolStack/mac/packet-scheduler/"
# symbol -> the reference header that declares it (the file:line the binding relies on is cited in integration/*.cpp)
REFERENCE_SYMBOLS = {
    # PacketScheduler and its per-TTI records (packet-scheduler.h:31, 50-137)
    "MAX_BEARERS": SCHED + "packet-scheduler.h",
    "UsersToSchedule": SCHED + "packet-scheduler.h",
    "UserToSchedule": SCHED + "packet-scheduler.h",
    "GetUsersToSchedule": SCHED + "packet-scheduler.h",
    "FlowsToSchedule": SCHED + "packet-scheduler.h",
    "FlowToSchedule": SCHED + "packet-scheduler.h",
    "GetFlowsToSchedule": SCHED + "packet-scheduler.h",
    "GetUserID": SCHED + "packet-scheduler.h",
    "GetUserNode": SCHED + "packet-scheduler.h",
    "GetCqiFeedbacks": SCHED + "packet-scheduler.h",
    "GetListOfAllocatedRBs": SCHED + "packet-scheduler.h",
    "UpdateAllocatedBits": SCHED + "packet-scheduler.h",
    "GetDataToTransmit": SCHED + "packet-scheduler.h",
    "GetBearer": SCHED + "packet-scheduler.h",
    "m_bearers": SCHED + "packet-scheduler.h",
    "m_dataToTransmit": SCHED + "packet-scheduler.h",
    "m_requiredRBs": SCHED + "packet-scheduler.h",
    "GetMacEntity": SCHED + "packet-scheduler.h",
    "SetMacEntity": SCHED + "packet-scheduler.h",
    "GetTimeStamp": SCHED + "packet-scheduler.h",
    "DoSchedule": SCHED + "packet-scheduler.h",
    # the three parents and the virtual each binding overrides
    "DownlinkTransportScheduler": SCHED + "downlink-transport-scheduler.h",
    "DownlinkNVSScheduler": SCHED + "downlink-nvs-scheduler.h",
    "DL_PF_PacketScheduler": SCHED + "dl-pf-packet-scheduler.h",
    # bearers
    "GetAverageTransmissionRate": "flows/radio-bearer.h",
    "GetHeadOfLinePacketDelay": "flows/radio-bearer.h",
    "GetDestination": "flows/radio-bearer-instance.h",
    # PDCCH message and the way to the PRB grid
    "PdcchMapIdealControlMessage": "core/idealMessages/ideal-control-messages.h",
    "AddNewRecord": "core/idealMessages/ideal-control-messages.h",
    "GetMessage": "core/idealMessages/ideal-control-messages.h",
    "DOWNLINK": "core/idealMessages/ideal-control-messages.h",
    "SendIdealControlMessage": "phy/lte-phy.h",
    "GetBandwidthManager": "phy/lte-phy.h",
    "GetDlSubChannels": "core/spectrum/bandwidth-manager.h",
    "GetDevice": "protocolStack/mac/mac-entity.h",
    "GetPhy": "device/NetworkNode.h",
    # the installation point
    "SetDLScheduler": "device/ENodeB.h",
    "SetDownlinkPacketScheduler": "protocolStack/mac/enb-mac-entity.h",
}
# identifiers after -> . :: that are NOT the reference's: the C++ library, jsoncpp (third party), this repository's own classes
NOT_REFERENCE = {
    "data", "size", "at", "push_back", "assign", "resize", "back", "vector", "string", "runtime_error", "cout", "endl", "ifstream",
    "is_open", "close", "h", "cpp", "inc", "so", "g",  # file-name pieces in comments
    "Value", "Reader", "parse", "asInt", "asDouble",  # jsoncpp: downlink-transport-scheduler.cpp:57-88 uses the same calls
    "LazyCreate", "DownlinkGpuScheduler", "DownlinkGpuNVSScheduler", "DL_GPU_PF_PacketScheduler", "RBsAllocation",
}


def _binding_text():
    return "\n".join(p.read_text() for p in sorted(INTEG.glob("*")) if p.suffix in (".h", ".cpp", ".inc"))


def _own_abi_identifiers():
    return set(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", (ROOT / "include" / "radiosaber_hip.h").read_text()))


def test_every_member_the_bindings_touch_is_accounted_for():
    """No identifier reached through -> . :: escapes the ledger: it is the reference's (and then checked below), the C ABI's, or
    on the short list of library names.  A new call in integration/ must be added to REFERENCE_SYMBOLS to pass."""
    used = set(re.findall(r"(?:->|\.|::)([A-Za-z_][A-Za-z0-9_]*)", _binding_text()))
    own = _own_abi_identifiers()
    gpu_enum = {u for u in used if u.startswith("DLScheduler_GPU_")}  # added by enodeb-cases.inc itself
    unknown = used - own - NOT_REFERENCE - set(REFERENCE_SYMBOLS) - gpu_enum
    assert not unknown, f"integration/ uses identifiers nobody vouches for: {sorted(unknown)}"
    unused = {s for s in REFERENCE_SYMBOLS if not re.search(rf"\b{s}\b", _binding_text())}
    assert not unused, f"ledger rows the bindings no longer use: {sorted(unused)}"


@pytest.mark.parametrize("symbol,header", sorted(REFERENCE_SYMBOLS.items()))
def test_reference_symbol_still_exists_with_that_spelling(symbol, header):
    text = (SRC / header).read_text(errors="replace")
    assert re.search(rf"\b{symbol}\b", text), f"{symbol} is gone from src/{header}: integration/ would no longer compile"


def test_signatures_the_bindings_depend_on():
    ps = (SRC / SCHED / "packet-scheduler.h").read_text()
    # public data members of struct UserToSchedule, arrays of MAX_BEARERS (integration reads m_bearers[b], m_dataToTransmit[p])
    assert re.search(r"RadioBearer\s*\*\s*m_bearers\s*\[\s*MAX_BEARERS\s*\]", ps)
    assert re.search(r"int\s+m_dataToTransmit\s*\[\s*MAX_BEARERS\s*\]", ps)
    assert re.search(r"const\s+int\s+MAX_BEARERS\s*=\s*2\s*;", ps)
    assert re.search(r"std::vector<int>\s*&?\s*GetCqiFeedbacks", ps) or re.search(r"GetCqiFeedbacks\s*\(", ps)
    # RBsAllocation is virtual in every parent, so the override is reached through DoSchedule()
    for h in ("downlink-transport-scheduler.h", "downlink-packet-scheduler.h", "downlink-nvs-scheduler.h"):
        assert re.search(r"virtual\s+void\s+RBsAllocation\s*\(", (SRC / SCHED / h).read_text()), h
    # the PDCCH record: (direction, PRB, destination node, MCS)
    icm = (SRC / "core/idealMessages/ideal-control-messages.h").read_text()
    assert re.search(r"AddNewRecord\s*\(\s*Direction\s+\w+\s*,\s*int\s+\w+\s*,\s*NetworkNode\s*\*\s*\w+\s*,\s*double\s+\w+\s*\)", icm)
    # the parents' constructors as the bindings call them
    assert re.search(r"DownlinkTransportScheduler\s*\(\s*std::string\s+\w+\s*,\s*int\s+\w+\s*\)",
                     (SRC / SCHED / "downlink-transport-scheduler.h").read_text())
    assert re.search(r"DownlinkNVSScheduler\s*\(\s*std::string\s+\w+(\s*=\s*\"\")?\s*,\s*bool\s+\w+",
                     (SRC / SCHED / "downlink-nvs-scheduler.h").read_text())
    assert re.search(r"DL_PF_PacketScheduler\s*\(\s*std::string\s*\w*\s*\)", (SRC / SCHED / "dl-pf-packet-scheduler.h").read_text())


def test_enodeb_edits_still_apply(tmp_path):
    """tools/check_integration.sh --patch-only: the enum entries, the three includes, the eight cases and the CLI numbers land in a
    scratch copy of the reference's files (the anchors they hang on still exist)."""
    r = subprocess.run([str(ROOT / "tools" / "check_integration.sh"), "--patch-only", str(REF), str(tmp_path)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    h = (tmp_path / "src/device/ENodeB.h").read_text()
    c = (tmp_path / "src/device/ENodeB.cpp").read_text()
    s = (tmp_path / "src/scenarios/single-cell-with-interference.h").read_text()
    enum = re.search(r"enum\s+DLSchedulerType\s*\{([^}]*)\}", h).group(1)
    assert enum.count("DLScheduler_GPU_") == 8 and "DLScheduler_VOGEL," in enum
    body = c[c.index("ENodeB::SetDLScheduler"):c.index("ENodeB::SetULScheduler")]
    assert body.count("case ENodeB::DLScheduler_GPU_") == 8
    assert body.index("DLScheduler_GPU_VOGEL") < body.rindex("default:")
    assert c.count("downlink-gpu-scheduler.h") == 1
    assert re.search(r"case 29:\s*downlink_scheduler_type = ENodeB::DLScheduler_GPU_MAXCELL;", s)
    # the reference tree itself was not touched
    assert "DLScheduler_GPU_" not in (SRC / "device/ENodeB.h").read_text()


def test_compile_check_refuses_without_jsoncpp_instead_of_faking_it():
    probe = subprocess.run("echo '#include <jsoncpp/json/json.h>' | g++ -x c++ -fsyntax-only -", shell=True, capture_output=True)
    if probe.returncode == 0:
        pytest.skip("jsoncpp is installed here: run tools/check_integration.sh /root/reference for the real check")
    r = subprocess.run([str(ROOT / "tools" / "check_integration.sh"), str(REF)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 77 and "no stand-ins" in r.stderr
