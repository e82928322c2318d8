"""The cache hints the kernels are designed around must be IN THE ISA. A run-time `if (flag) nontemporal_load else load`
is merged by the compiler into one plain load (the hint does not survive the merge): the quantise and pool kernels ran
without a single non-temporal access for most of two rounds while their source, their tunables and their documentation
said otherwise. This test compiles the device code of the two files to assembly (hipcc cross-compiles, no GPU) and
looks at the instructions of the shipped instantiations."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "efficient-llm-inference_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _asm(tmp_path_factory, name):
    if not shutil.which(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / (name + ".s")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                    "--offload-device-only", "-S", os.path.join(CSRC, name + ".hip"), "-o", str(out)],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    return out.read_text()


def _kernel_body(asm, symbol_regex):
    """Instructions of the first kernel whose mangled name matches."""
    m = re.search(r"^(%s):[^\n]*\n(.*?)\n\s*s_endpgm" % symbol_regex, asm, flags=re.M | re.S)
    assert m, "kernel not found: " + symbol_regex
    return m.group(2)


@pytest.fixture(scope="module")
def quant_asm(tmp_path_factory):
    return _asm(tmp_path_factory, "kvq_quant")


@pytest.fixture(scope="module")
def evict_asm(tmp_path_factory):
    return _asm(tmp_path_factory, "kvq_evict")


@pytest.mark.parametrize("bits", [4, 8])
@pytest.mark.parametrize("geo", [(8, 16, 4), (8, 8, 8), (12, 8, 8), (16, 8, 8)])  # R rows, D/8, tokens per tile
def test_shipped_quantise_tile_issues_non_temporal_accesses(quant_asm, bits, geo):
    # quant_tile_k<f16, BITS, R, DV, TT, PHASE = 0>: R buffer loads of 16 B per lane, every one non-temporal; every
    # LDS-staged 16-byte output store non-temporal (the 4-byte scale store is a plain global store)
    r, dv, tt = geo
    body = _kernel_body(quant_asm, r"_ZN3kvq12quant_tile_kILi0ELi%dELi%dELi%dELi%dELi0EEEvNS_13QuantTileArgsE" % (bits, r, dv, tt))
    loads = re.findall(r"buffer_load_dwordx4[^\n]*", body)
    assert len(loads) == r and all(l.rstrip().endswith(" nt") for l in loads), loads
    stores = re.findall(r"buffer_store_dwordx4[^\n]*", body)
    assert len(stores) == (r * tt * dv * bits + 1023) // 1024 and all(s.rstrip().endswith(" nt") for s in stores), stores


def test_split_phase_tiles_keep_phase_one_rows_cacheable(quant_asm):
    # the abs-max phase of a batch-sharded slice reads with PLAIN loads (its rows are re-read by the quantise phase from
    # the Infinity Cache), the quantise phase with non-temporal ones (last use)
    p1 = _kernel_body(quant_asm, r"_ZN3kvq12quant_tile_kILi0ELi8ELi8ELi16ELi4ELi1EEEvNS_13QuantTileArgsE")
    p2 = _kernel_body(quant_asm, r"_ZN3kvq12quant_tile_kILi0ELi4ELi8ELi16ELi4ELi2EEEvNS_13QuantTileArgsE")
    l1 = re.findall(r"buffer_load_dwordx4[^\n]*", p1)
    l2 = re.findall(r"buffer_load_dwordx4[^\n]*", p2)
    assert len(l1) == 8 and not any(l.rstrip().endswith(" nt") for l in l1), l1
    assert len(l2) == 8 and all(l.rstrip().endswith(" nt") for l in l2), l2
    assert not re.findall(r"buffer_store", p1) and re.findall(r"global_atomic_umax", p1)


@pytest.mark.parametrize("bits", [4, 8])
def test_wide_quantise_tile_holds_its_slice_in_registers(quant_asm, bits):
    # quant_wide_k<f16, BITS, 1024, 16>: sixteen 16-byte non-temporal loads per lane, sixteen non-temporal stores of the
    # packed vector (4 / 8 bytes), no scratch: the 64 data VGPRs + arithmetic fit the 128 a 1024-thread workgroup's lanes have
    sym = r"_ZN3kvq12quant_wide_kILi0ELi%dELi1024ELi16EEEvNS_13QuantWideArgsE" % bits
    body = _kernel_body(quant_asm, sym)
    loads = re.findall(r"global_load_dwordx4[^\n]*", body)
    assert len(loads) == 16 and all(l.rstrip().endswith(" nt") for l in loads), loads
    stores = re.findall(r"global_store_dword%s [^\n]*" % ("x2" if bits == 8 else ""), body)
    nt = [s for s in stores if s.rstrip().endswith(" nt")]  # (INT4: the one plain dword store is the per-token scale)
    assert len(nt) == 16 and len(stores) - len(nt) <= 1, stores
    assert "scratch_" not in body
    m = re.search(r"\.amdhsa_kernel %s\b.*?\.amdhsa_next_free_vgpr (\d+)" % sym, quant_asm, flags=re.S)
    assert m and int(m.group(1)) <= 128, m and m.group(1)


def test_pool_kernels_issue_non_temporal_loads(evict_asm):
    wave = _kernel_body(evict_asm, r"_ZN3kvq17chunk_pool_wave_kILi0ELi4ELi16EEEvNS_8PoolArgsE")
    loads = re.findall(r"buffer_load_dwordx4[^\n]*", wave)
    assert len(loads) == 16 and all(l.rstrip().endswith(" nt") for l in loads), loads
    vec_nt = _kernel_body(evict_asm, r"_ZN3kvq16chunk_pool_vec_kILi0ELb1EEEvNS_8PoolArgsEj")
    # the unrolled batch of 16 loads carries the hint (the tail loop's single load is a plain one; the plain-load
    # instantiation exists in A-B builds only)
    assert len(re.findall(r"global_load_dwordx4[^\n]* nt\n", vec_nt + "\n")) >= 16


def test_shipped_dequantise_variants_store_non_temporally(tmp_path_factory):
    asm = _asm(tmp_path_factory, "kvq_dequant")
    # variant 21 (INT4): dequant_tokens_fast_k<f16, 4, LE 8, UNROLL 4, NT stores, LDS scales, plain loads, 64 threads>
    i4 = _kernel_body(asm, r"_ZN3kvq21dequant_tokens_fast_kILi0ELi4ELi8ELi4ELb1ELb1ELb0ELi64EEEvNS_11DequantArgsE")
    st = re.findall(r"global_store_dwordx4[^\n]*", i4)
    assert st and all(s.rstrip().endswith(" nt") for s in st), st
    # variant 23 (INT8): UNROLL 2, NT stores AND NT loads of the quantised bytes
    i8 = _kernel_body(asm, r"_ZN3kvq21dequant_tokens_fast_kILi0ELi8ELi8ELi2ELb1ELb1ELb1ELi64EEEvNS_11DequantArgsE")
    st = re.findall(r"global_store_dwordx4[^\n]*", i8)
    assert st and all(s.rstrip().endswith(" nt") for s in st), st
    ld = re.findall(r"global_load_dwordx2[^\n]*", i8)
    assert ld and all(l.rstrip().endswith(" nt") for l in ld), ld
