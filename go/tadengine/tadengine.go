// Package tadengine is the cgo binding of libtad_mi355x.so (include/tad.h) for theia-manager.
//
// STATUS: written against include/tad.h but NOT compiled or tested here — this build image has no
// Go toolchain (`go version`: not found).  The identical C ABI is exercised by the ctypes binding
// (theia_amd/_capi.py), tests/test_capi_abi.py and the C driver tools/capi_driver.c.
//
// It replaces the SparkApplication launch of
// pkg/controller/anomalydetector/controller.go:525-698 (startSparkApplication) with an in-process
// call: the controller reads the job's columns from ClickHouse, dictionary-encodes the key columns,
// calls Engine.Run, and inserts the returned rows into default.tadetector.
package tadengine

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../theia_amd/lib -ltad_mi355x -Wl,-rpath,${SRCDIR}/../../theia_amd/lib
#include <stdlib.h>
#include <string.h>
#include "tad.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"runtime"
	"unsafe"
)

// KeySkip marks a row (or its second key) that the SQL predicates reject.
const KeySkip = ^uint64(0)

type Algo int

const (
	EWMA   Algo = C.TAD_ALGO_EWMA
	ARIMA  Algo = C.TAD_ALGO_ARIMA
	DBSCAN Algo = C.TAD_ALGO_DBSCAN
	Drop   Algo = C.TAD_ALGO_DROP // abnormal-traffic-drop detector (snowflake/udfs/udfs/drop_detection)
)

type AggFlow int

const (
	AggNone     AggFlow = C.TAD_AGG_NONE
	AggPod      AggFlow = C.TAD_AGG_POD
	AggSvc      AggFlow = C.TAD_AGG_SVC
	AggExternal AggFlow = C.TAD_AGG_EXTERNAL
)

// IllegalArgument mirrors illeagelArguementError (controller.go:505-514): the job is marked FAILED
// and not retried.
type IllegalArgument struct{ Msg string }

func (e IllegalArgument) Error() string { return e.Msg }

// Engine owns one GPU.  Run may be called from the controller's 4 workers concurrently
// (controller.go:199-201): each call takes one of the engine's job contexts (own HIP stream and workspace, tad.h ABI 12), so up
// to MaxJobsInFlight jobs overlap on the GPU; further callers wait inside the library.
type Engine struct{ h *C.tad_engine }

// Plan mirrors tad_plan (tad.h, ABI 12): plan overrides of an engine, every field 0 = the engine decides — what the controller
// uses.  Tests and A/B measurements force a strategy with it; the library reads no environment variable.  A Go struct, not
// C.tad_plan: cgo types are private to this package, callers in other packages could not construct one.
type Plan struct {
	Stage0        int32  // 1 = direct atomic scatter, 2 = partition + LDS tiles whatever the batch size
	PartitionPass int32  // 1 = sort-by-tile pass B, 2 = write-combining pass B, 3 = write-combining with 64-byte sectors only
	Histogram     int32  // 1 = exact per-workgroup histogram in pass A
	Sparse        int32  // 1 = never, 2 = always the sort-based Stage 0 for sparse tables
	SparseClasses int32  // 1 = always run a sparse table as length classes of keys
	EwmaEmit      int32  // 1 = lane-per-key emit for the EWMA job
	EwmaEmitRows  uint32 // LDS rows per wavefront of the staged EWMA emit (<= 4096)
	TileCells     int32  // 1 = 8-byte tile cells in the settle mode of DBSCAN jobs with max
	SparseSort    int32  // sparse tables: 1 = always the LSD radix sort, 2 = the partition pass + LDS sort wherever its plan fits (ABI 9)
}

func (p Plan) c() C.tad_plan {
	return C.tad_plan{stage0: C.int32_t(p.Stage0), partition_pass: C.int32_t(p.PartitionPass), histogram: C.int32_t(p.Histogram),
		sparse: C.int32_t(p.Sparse), sparse_classes: C.int32_t(p.SparseClasses), ewma_emit: C.int32_t(p.EwmaEmit),
		ewma_emit_rows: C.uint32_t(p.EwmaEmitRows), tile_cells: C.int32_t(p.TileCells), sparse_sort: C.int32_t(p.SparseSort)}
}

func NewEngine(device int) (*Engine, error) { return NewEngineWithPlan(device, Plan{}) }

// NewEngineWithPlan creates the engine with plan overrides in tad_engine_opts.plan and the default number of job contexts (4).
func NewEngineWithPlan(device int, plan Plan) (*Engine, error) { return NewEngineWithOptions(device, plan, 0) }

// NewEngineWithOptions: maxJobsInFlight = tad_engine_opts.max_jobs_in_flight (0 = 4, the controller's worker count; 1 = jobs serialise).
func NewEngineWithOptions(device int, plan Plan, maxJobsInFlight int) (*Engine, error) {
	// the header this file was compiled against and the library the process loaded must be the same ABI (struct layouts!)
	if v := int(C.tad_abi_version()); v != int(C.TAD_ABI_VERSION) {
		return nil, fmt.Errorf("libtad_mi355x.so ABI %d != tad.h ABI %d", v, int(C.TAD_ABI_VERSION))
	}
	opts := C.tad_engine_opts{device: C.int32_t(device), plan: plan.c(), max_jobs_in_flight: C.int32_t(maxJobsInFlight)}
	var h *C.tad_engine
	if rc := C.tad_engine_create(&opts, &h); rc != C.TAD_OK {
		return nil, fmt.Errorf("tad_engine_create: %s (code %d)", C.GoString(C.tad_last_error(nil)), int(rc))
	}
	e := &Engine{h: h}
	runtime.SetFinalizer(e, func(e *Engine) { e.Close() })
	return e, nil
}

// SetPlan replaces the engine's plan overrides; takes effect with the next job.
func (e *Engine) SetPlan(p Plan) error {
	cp := p.c()
	if rc := C.tad_engine_set_plan(e.h, &cp); rc != C.TAD_OK {
		return fmt.Errorf("tad_engine_set_plan: %s (code %d)", C.GoString(C.tad_last_error(e.h)), int(rc))
	}
	return nil
}

func (e *Engine) Close() {
	if e.h != nil {
		C.tad_engine_destroy(e.h)
		e.h = nil
	}
}

// Job mirrors the SparkApplication argument vector (controller.go:526-623).
type Job struct {
	Algo      Algo
	AggFlow   AggFlow
	StartTime int64 // epoch seconds, 0 = unset (Spec.StartInterval)
	EndTime   int64 // epoch seconds, 0 = unset (Spec.EndInterval)
	ID        string
}

// Columns is one batch of flow rows after dictionary encoding; all slices have the same length.
type Columns struct {
	KeyID       []uint64
	KeyID2      []uint64 // pod mode only (inbound/outbound UNION ALL), else nil
	FlowEndS    []int64
	FlowStartS  []int64 // nil unless StartTime is set
	Value       []uint64
	NumKeys     uint64
	KeyHist     *KeyHist // optional: FactorizeHist's by-product for THIS batch (Stage 0 then does not count the key column again)
}

// KeyHist is tad_key_hist (tad.h, ABI 12): the key-bin histogram of a factorised batch per Stage-0 workgroup.  The bins live in device
// memory (TAD_KEY_HIST_BYTES, allocated by FactorizeHist, released by Free); the struct itself is C memory so that tad_columns may point to it.
type KeyHist struct{ c *C.tad_key_hist }

func (h *KeyHist) Valid() bool { return h != nil && h.c != nil && h.c.n_rows != 0 }

func (h *KeyHist) Free(e *Engine) {
	if h != nil && h.c != nil {
		C.tad_device_free(e.h, unsafe.Pointer(h.c.bins))
		C.free(unsafe.Pointer(h.c))
		h.c = nil
	}
}

// Row is the mode-independent part of one tadetector row (create_table.sh:363-384).
type Row struct {
	KeyID      uint64
	FlowEndS   int64
	Throughput float64
	AlgoCalc   float64
	StdDev     float64
}

type Stats struct {
	RowsIn, RowsUsed, Keys, Points, Anomalies, KeysNoResult uint64
	ArimaNanFits                                             uint64 // ARIMA fits voided by a non-finite likelihood (tad.h: tad_stats.arima_nan_fits)
	MsTotal                                                  float32
	Stage0Path                                               int32 // how Stage 0 ran (tad.h: tad_stats.stage0_path), for the controller's logs
	HostSyncs                                                int32 // host synchronisations of the job (tad.h: tad_stats.host_syncs): 3, or 2 with a lattice hint
	JobContext                                               int32 // which of the engine's job contexts ran it (tad.h: tad_stats.job_context, ABI 12)
	ArimaRelaunches                                          int32 // times the ARIMA fit yielded to other jobs' whole-CU kernels and was relaunched (tad_stats.arima_relaunches)
}

// cColumn copies a Go slice into C memory: cgo forbids handing Go pointers nested in a C struct, and the
// library stages host columns to the GPU anyway.  A production binding would read ClickHouse blocks
// straight into C/pinned buffers instead of Go slices.
func cColumn[T uint64 | int64](s []T) unsafe.Pointer {
	if len(s) == 0 {
		return nil
	}
	n := C.size_t(len(s) * 8)
	p := C.malloc(n)
	C.memcpy(p, unsafe.Pointer(&s[0]), n)
	return p
}

// Run replaces one SparkApplication run (anomaly_detection.py:647-710).  An empty result means the caller
// writes the "NO ANOMALY DETECTED" sentinel row (anomaly_detection.py:395-420).
func (e *Engine) Run(job Job, cols Columns) ([]Row, Stats, error) {
	var st Stats
	n := len(cols.KeyID)
	if len(cols.FlowEndS) != n || len(cols.Value) != n || (cols.KeyID2 != nil && len(cols.KeyID2) != n) ||
		(cols.FlowStartS != nil && len(cols.FlowStartS) != n) {
		return nil, st, errors.New("tadengine: columns differ in length")
	}
	var cj C.tad_job
	cj.algo = C.tad_algo(job.Algo)
	cj.agg_flow = C.tad_agg_flow(job.AggFlow)
	cj.value_op = C.TAD_OP_AUTO
	cj.start_time = C.int64_t(job.StartTime)
	cj.end_time = C.int64_t(job.EndTime)
	id := []byte(job.ID)
	if len(id) > 63 {
		id = id[:63]
	}
	for i, b := range id {
		cj.id[i] = C.char(b)
	}
	var cc C.tad_columns
	cc.n_rows = C.uint64_t(n)
	cc.num_keys = C.uint64_t(cols.NumKeys)
	cc.memory = C.TAD_MEM_HOST
	bufs := []unsafe.Pointer{cColumn(cols.KeyID), cColumn(cols.KeyID2), cColumn(cols.FlowEndS), cColumn(cols.FlowStartS), cColumn(cols.Value)}
	defer func() {
		for _, p := range bufs {
			if p != nil {
				C.free(p)
			}
		}
	}()
	cc.key_id = (*C.uint64_t)(bufs[0])
	cc.key_id2 = (*C.uint64_t)(bufs[1])
	cc.flow_end_s = (*C.int64_t)(bufs[2])
	cc.flow_start_s = (*C.int64_t)(bufs[3])
	cc.value = (*C.uint64_t)(bufs[4])
	if cols.KeyHist.Valid() {
		cc.key_hist = cols.KeyHist.c // C memory: no Go pointer inside the struct handed to C
	}

	var res *C.tad_result
	rc := C.tad_run(e.h, &cj, &cc, C.TAD_MEM_HOST, &res)
	if rc != C.TAD_OK {
		msg := C.GoString(C.tad_last_error(e.h))
		if rc == C.TAD_ERR_INVALID_ARGUMENT {
			return nil, st, IllegalArgument{msg}
		}
		return nil, st, fmt.Errorf("tad_run: %s (code %d)", msg, int(rc))
	}
	defer C.tad_result_free(e.h, res)
	a := int(res.n_rows)
	rows := make([]Row, a)
	if a > 0 {
		k := unsafe.Slice((*uint64)(unsafe.Pointer(res.key_id)), a)
		t := unsafe.Slice((*int64)(unsafe.Pointer(res.flow_end_s)), a)
		x := unsafe.Slice((*float64)(unsafe.Pointer(res.throughput)), a)
		c := unsafe.Slice((*float64)(unsafe.Pointer(res.algo_calc)), a)
		s := unsafe.Slice((*float64)(unsafe.Pointer(res.stddev)), a)
		for i := range rows {
			rows[i] = Row{k[i], t[i], x[i], c[i], s[i]}
		}
	}
	st = Stats{uint64(res.stats.rows_in), uint64(res.stats.rows_used), uint64(res.stats.n_keys), uint64(res.stats.n_points),
		uint64(res.stats.n_anomalies), uint64(res.stats.keys_no_result), uint64(res.stats.arima_nan_fits), float32(res.stats.ms_total),
		int32(res.stats.stage0_path), int32(res.stats.host_syncs), int32(res.stats.job_context), int32(res.stats.arima_relaunches)}
	return rows, st, nil
}

// Point is one aggregated (key, flowEndSeconds) point of Stage 0 (tad_aggregate): the GROUP BY the reference pushes
// into ClickHouse, with the full UInt64 aggregate.  Row-sharded ingest across several GPUs aggregates each slice
// with Aggregate, routes the points to the engine that owns the key, and calls Run on the partial points.
type Point struct {
	KeyID    uint64
	FlowEndS int64
	Value    uint64
}

func (e *Engine) Aggregate(job Job, cols Columns) ([]Point, error) {
	var cj C.tad_job
	cj.agg_flow = C.tad_agg_flow(job.AggFlow)
	cj.value_op = C.TAD_OP_AUTO
	cj.start_time = C.int64_t(job.StartTime)
	cj.end_time = C.int64_t(job.EndTime)
	var cc C.tad_columns
	n := len(cols.KeyID)
	cc.n_rows = C.uint64_t(n)
	cc.num_keys = C.uint64_t(cols.NumKeys)
	cc.memory = C.TAD_MEM_HOST
	bufs := []unsafe.Pointer{cColumn(cols.KeyID), cColumn(cols.KeyID2), cColumn(cols.FlowEndS), cColumn(cols.FlowStartS), cColumn(cols.Value)}
	defer func() {
		for _, p := range bufs {
			if p != nil {
				C.free(p)
			}
		}
	}()
	cc.key_id = (*C.uint64_t)(bufs[0])
	cc.key_id2 = (*C.uint64_t)(bufs[1])
	cc.flow_end_s = (*C.int64_t)(bufs[2])
	cc.flow_start_s = (*C.int64_t)(bufs[3])
	cc.value = (*C.uint64_t)(bufs[4])
	var pts *C.tad_points
	if rc := C.tad_aggregate(e.h, &cj, &cc, C.TAD_MEM_HOST, &pts); rc != C.TAD_OK {
		return nil, fmt.Errorf("tad_aggregate: %s (code %d)", C.GoString(C.tad_last_error(e.h)), int(rc))
	}
	defer C.tad_points_free(e.h, pts)
	m := int(pts.n_points)
	out := make([]Point, m)
	if m > 0 {
		k := unsafe.Slice((*uint64)(unsafe.Pointer(pts.key_id)), m)
		t := unsafe.Slice((*int64)(unsafe.Pointer(pts.flow_end_s)), m)
		v := unsafe.Slice((*uint64)(unsafe.Pointer(pts.value)), m)
		for i := range out {
			out[i] = Point{k[i], t[i], v[i]}
		}
	}
	return out, nil
}

// ShardRows buckets device-resident rows by the owner of their key (owner = key mod world, local id = key / world) for
// the all-to-all(v) of row-sharded multi-GPU ingest (tad_shard_rows).  key, flowEnd, value and the three outputs are
// DEVICE pointers to n 8-byte elements each (AllocDevice); the returned counts are the rows per destination rank.
func (e *Engine) ShardRows(key, flowEnd, value unsafe.Pointer, n uint64, world uint32, outKey, outFlowEnd, outValue unsafe.Pointer) ([]uint64, error) {
	var cc C.tad_columns
	cc.n_rows = C.uint64_t(n)
	cc.memory = C.TAD_MEM_DEVICE
	cc.key_id = (*C.uint64_t)(key)
	cc.flow_end_s = (*C.int64_t)(flowEnd)
	cc.value = (*C.uint64_t)(value)
	if world == 0 {
		return nil, fmt.Errorf("tad_shard_rows: world must be >= 1")
	}
	counts := make([]uint64, world)
	if rc := C.tad_shard_rows(e.h, &cc, C.uint32_t(world), (*C.uint64_t)(outKey), (*C.int64_t)(outFlowEnd), (*C.uint64_t)(outValue),
		(*C.uint64_t)(unsafe.Pointer(&counts[0]))); rc != C.TAD_OK {
		return nil, fmt.Errorf("tad_shard_rows: %s (code %d)", C.GoString(C.tad_last_error(e.h)), int(rc))
	}
	return counts, nil
}

// Factorize turns the rows' GROUP BY key tuples — dictionary codes of the string columns, ports, protocol, flowStartSeconds: up to
// eight int64 columns (anomaly_detection.py:52-137) — into dense key ids in order of first appearance on the GPU (tad_factorize).
// keep (nil = every row) marks the rows the SQL's predicates accept; the others get KeySkip.  colsB / keepB: the second tuple of
// every row in pod mode (the outbound view of the UNION ALL, :556-565), ids into the second return value.  firstRow[k] = the virtual
// row (i for side a, n + i for side b) where key k first appears: the caller reads the key's column values there.
func (e *Engine) Factorize(colsA [][]int64, keepA []byte, colsB [][]int64, keepB []byte) (keyID, keyID2, firstRow []uint64, err error) {
	keyID, keyID2, firstRow, _, err = e.factorize(colsA, keepA, colsB, keepB, false)
	return
}

// FactorizeHist is Factorize plus the key-bin histogram of the ids (tad_factorize_hist): put it into Columns.KeyHist of the job over the
// same rows and Stage 0 sizes its partition regions from it instead of reading the key column a second time.  Free it after the job.
func (e *Engine) FactorizeHist(colsA [][]int64, keepA []byte, colsB [][]int64, keepB []byte) (keyID, keyID2, firstRow []uint64, hist *KeyHist, err error) {
	return e.factorize(colsA, keepA, colsB, keepB, true)
}

func (e *Engine) factorize(colsA [][]int64, keepA []byte, colsB [][]int64, keepB []byte, withHist bool) (keyID, keyID2, firstRow []uint64, hist *KeyHist, err error) {
	if len(colsA) < 1 || len(colsA) > 8 || (colsB != nil && len(colsB) != len(colsA)) {
		return nil, nil, nil, nil, errors.New("tadengine: 1..8 key columns, the same number on both sides")
	}
	n := len(colsA[0])
	sides := 1
	if colsB != nil {
		sides = 2
	}
	var bufs []unsafe.Pointer
	defer func() {
		for _, p := range bufs {
			if p != nil {
				C.free(p)
			}
		}
	}()
	ptrs := func(cols [][]int64) *unsafe.Pointer {
		arr := (*[8]unsafe.Pointer)(C.malloc(C.size_t(8 * unsafe.Sizeof(unsafe.Pointer(nil)))))
		bufs = append(bufs, unsafe.Pointer(arr))
		for c, col := range cols {
			if len(col) != n {
				return nil
			}
			arr[c] = cColumn(col)
			bufs = append(bufs, arr[c])
		}
		return &arr[0]
	}
	mask := func(m []byte) unsafe.Pointer {
		if m == nil {
			return nil
		}
		p := C.CBytes(m)
		bufs = append(bufs, p)
		return p
	}
	var kc C.tad_key_columns
	kc.n_rows = C.uint64_t(n)
	kc.n_cols = C.int32_t(len(colsA))
	kc.memory = C.TAD_MEM_HOST
	pa := ptrs(colsA)
	if pa == nil || (keepA != nil && len(keepA) != n) || (keepB != nil && len(keepB) != n) {
		return nil, nil, nil, nil, errors.New("tadengine: key columns and masks differ in length")
	}
	kc.cols_a = (**C.int64_t)(unsafe.Pointer(pa))
	kc.keep_a = (*C.uint8_t)(mask(keepA))
	if colsB != nil {
		pb := ptrs(colsB)
		if pb == nil {
			return nil, nil, nil, nil, errors.New("tadengine: key columns differ in length")
		}
		kc.cols_b = (**C.int64_t)(unsafe.Pointer(pb))
		kc.keep_b = (*C.uint8_t)(mask(keepB))
	}
	keyID = make([]uint64, n)
	firstRow = make([]uint64, n*sides)
	var k2 *C.uint64_t
	if colsB != nil {
		keyID2 = make([]uint64, n)
		if n > 0 {
			k2 = (*C.uint64_t)(unsafe.Pointer(&keyID2[0]))
		}
	}
	if n == 0 {
		return keyID, keyID2, firstRow[:0], nil, nil
	}
	var nk C.uint64_t
	if withHist {
		hist = &KeyHist{c: (*C.tad_key_hist)(C.calloc(1, C.size_t(unsafe.Sizeof(C.tad_key_hist{}))))}
		var bins unsafe.Pointer
		if rc := C.tad_device_alloc(e.h, C.uint64_t(C.TAD_KEY_HIST_BYTES), &bins); rc != C.TAD_OK {
			C.free(unsafe.Pointer(hist.c))
			return nil, nil, nil, nil, fmt.Errorf("tad_device_alloc: %s (code %d)", C.GoString(C.tad_last_error(e.h)), int(rc))
		}
		hist.c.bins = (*C.uint32_t)(bins)
		if rc := C.tad_factorize_hist(e.h, &kc, (*C.uint64_t)(unsafe.Pointer(&keyID[0])), k2, (*C.uint64_t)(unsafe.Pointer(&firstRow[0])),
			C.uint64_t(len(firstRow)), &nk, hist.c); rc != C.TAD_OK {
			hist.Free(e)
			return nil, nil, nil, nil, fmt.Errorf("tad_factorize_hist: %s (code %d)", C.GoString(C.tad_last_error(e.h)), int(rc))
		}
		return keyID, keyID2, firstRow[:int(nk)], hist, nil
	}
	if rc := C.tad_factorize(e.h, &kc, (*C.uint64_t)(unsafe.Pointer(&keyID[0])), k2, (*C.uint64_t)(unsafe.Pointer(&firstRow[0])),
		C.uint64_t(len(firstRow)), &nk); rc != C.TAD_OK {
		return nil, nil, nil, nil, fmt.Errorf("tad_factorize: %s (code %d)", C.GoString(C.tad_last_error(e.h)), int(rc))
	}
	return keyID, keyID2, firstRow[:int(nk)], nil, nil
}


// EncodeStrings turns one string column of a batch — in Arrow's layout, what clickhouse-go's column-oriented block API and the
// Arrow Go reader both hand out: n+1 offsets into a byte slice — into dictionary codes on the GPU (tad.h: tad_encode_strings, ABI 10):
// codes[i] = id of row i's string in order of first appearance, firstRow[k] = the row where value k first appears (the host reads
// the dictionary's strings there, evaluates the job's string predicates on them and passes the codes on to Factorize).
// validity may be nil (no nulls); a null row encodes like "".
func (e *Engine) EncodeStrings(offsets []int32, data []byte, validity []byte) (codes []int64, firstRow []uint64, err error) {
	if len(offsets) == 0 {
		return nil, nil, errors.New("tadengine: offsets hold n + 1 entries")
	}
	n := len(offsets) - 1
	codes = make([]int64, n)
	firstRow = make([]uint64, n)
	if n == 0 {
		return codes, firstRow, nil
	}
	// sc is Go memory that HOLDS Go pointers (offsets, data, validity).  cgo only accepts that when the pointed-to memory is pinned
	// ("cgo argument has Go pointer to unpinned Go pointer" otherwise, with the default cgocheck): the three slices are pinned for
	// the duration of the call (runtime.Pinner, Go >= 1.21) — the column's bytes stay zero-copy, unlike Factorize's small id columns,
	// which are copied into C memory.  codes and firstRow are passed directly and hold no pointers.
	var pin runtime.Pinner
	defer pin.Unpin()
	var sc C.tad_string_column
	sc.n_rows = C.uint64_t(n)
	pin.Pin(&offsets[0])
	sc.offsets = unsafe.Pointer(&offsets[0])
	sc.offset_bits = 32
	if len(data) > 0 {
		pin.Pin(&data[0])
		sc.data = (*C.uint8_t)(unsafe.Pointer(&data[0]))
	}
	sc.data_bytes = C.uint64_t(len(data))
	if validity != nil {
		if len(validity)*8 < n {
			return nil, nil, errors.New("tadengine: validity bitmap shorter than the column")
		}
		pin.Pin(&validity[0])
		sc.validity = (*C.uint8_t)(unsafe.Pointer(&validity[0]))
	}
	sc.memory = C.TAD_MEM_HOST
	var nv C.uint64_t
	if rc := C.tad_encode_strings(e.h, &sc, (*C.int64_t)(unsafe.Pointer(&codes[0])), (*C.uint64_t)(unsafe.Pointer(&firstRow[0])),
		C.uint64_t(len(firstRow)), &nv); rc != C.TAD_OK {
		return nil, nil, fmt.Errorf("tad_encode_strings: %s (code %d)", C.GoString(C.tad_last_error(e.h)), int(rc))
	}
	return codes, firstRow[:int(nv)], nil
}

// AllocDevice / FreeDevice / CopyToDevice / CopyToHost: device buffers for hosts without a HIP binding of their own (what ShardRows and a
// device-resident Run take).
// ---- columnar ingest (tad.h ABI 12): Arrow buffers -> the engine's 8-byte device columns ----

// AllocHost returns page-locked host memory (tad_host_alloc): a reader receives ClickHouse's ArrowStream body straight into it and copies from
// it to the device run at PCIe rate.  Keep and reuse the buffers between jobs: pinning is slow.
func (e *Engine) AllocHost(bytes uint64) (unsafe.Pointer, error) {
	var p unsafe.Pointer
	if rc := C.tad_host_alloc(e.h, C.uint64_t(bytes), &p); rc != C.TAD_OK {
		return nil, fmt.Errorf("tad_host_alloc: %s (code %d)", C.GoString(C.tad_last_error(e.h)), int(rc))
	}
	return p, nil
}

func (e *Engine) FreeHost(p unsafe.Pointer) { C.tad_host_free(e.h, p) }

// WidenColumn writes dst[i] = table[src[i]] (table != nil: the dictionary indices of one Arrow record batch through the batch's remap into the
// column's job-wide dictionary; or a gather of a device column at Factorize's first rows) or src[i] widened (UInt32 DateTime, UInt16 ports)
// for i < n.  src: n integers of srcBits (8 / 16 / 32 / 64) bits in C / page-locked host memory (srcOnDevice false) or device memory; table and
// dst: device memory (int64).  Pointers are plain C pointers: nothing of Go's heap crosses the boundary.
func (e *Engine) WidenColumn(src unsafe.Pointer, srcBits int, srcSigned, srcOnDevice bool, n uint64, table unsafe.Pointer, tableLen uint64, dst unsafe.Pointer) error {
	signed, mem := C.int32_t(0), C.tad_mem(C.TAD_MEM_HOST)
	if srcSigned {
		signed = 1
	}
	if srcOnDevice {
		mem = C.TAD_MEM_DEVICE
	}
	if rc := C.tad_widen_column(e.h, src, C.int32_t(srcBits), signed, mem, C.uint64_t(n), (*C.int64_t)(table), C.uint64_t(tableLen), (*C.int64_t)(dst)); rc != C.TAD_OK {
		return fmt.Errorf("tad_widen_column: %s (code %d)", C.GoString(C.tad_last_error(e.h)), int(rc))
	}
	return nil
}

// MaskRows writes keep[i] = AND over t of masks[t][codes[t][i]] (ANDed into the previous keep[i] when combine): the SQL's string predicates
// (anomaly_detection.py:507-614), evaluated by the host on the DISTINCT values of each column, applied to the rows on the GPU.  codes[t]
// (int64[n]), masks[t] (uint8[maskLen[t]]) and keep (uint8[n]) are device pointers; at most 8 terms.
func (e *Engine) MaskRows(n uint64, codes, masks []unsafe.Pointer, maskLen []uint64, combine bool, keep unsafe.Pointer) error {
	k := len(codes)
	if k > 8 || len(masks) != k || len(maskLen) != k {
		return errors.New("tadengine: MaskRows takes at most 8 terms, one mask and one length per code column")
	}
	// the two pointer arrays live in C memory for the call (cgo may not pass Go memory that holds pointers)
	carr := (*[8]unsafe.Pointer)(C.malloc(C.size_t(8 * unsafe.Sizeof(unsafe.Pointer(nil)))))
	marr := (*[8]unsafe.Pointer)(C.malloc(C.size_t(8 * unsafe.Sizeof(unsafe.Pointer(nil)))))
	larr := (*[8]C.uint64_t)(C.malloc(C.size_t(8 * 8)))
	defer C.free(unsafe.Pointer(carr))
	defer C.free(unsafe.Pointer(marr))
	defer C.free(unsafe.Pointer(larr))
	for t := 0; t < k; t++ {
		carr[t], marr[t], larr[t] = codes[t], masks[t], C.uint64_t(maskLen[t])
	}
	comb := C.int32_t(0)
	if combine {
		comb = 1
	}
	if rc := C.tad_mask_rows(e.h, C.uint64_t(n), C.int32_t(k), (**C.int64_t)(unsafe.Pointer(&carr[0])), (**C.uint8_t)(unsafe.Pointer(&marr[0])),
		&larr[0], comb, (*C.uint8_t)(keep)); rc != C.TAD_OK {
		return fmt.Errorf("tad_mask_rows: %s (code %d)", C.GoString(C.tad_last_error(e.h)), int(rc))
	}
	return nil
}

func (e *Engine) AllocDevice(bytes uint64) (unsafe.Pointer, error) {
	var p unsafe.Pointer
	if rc := C.tad_device_alloc(e.h, C.uint64_t(bytes), &p); rc != C.TAD_OK {
		return nil, fmt.Errorf("tad_device_alloc: %s (code %d)", C.GoString(C.tad_last_error(e.h)), int(rc))
	}
	return p, nil
}

func (e *Engine) FreeDevice(p unsafe.Pointer) { C.tad_device_free(e.h, p) }

func (e *Engine) CopyToDevice(dst unsafe.Pointer, src []byte) error {
	if len(src) == 0 {
		return nil
	}
	if rc := C.tad_copy_to_device(e.h, dst, unsafe.Pointer(&src[0]), C.uint64_t(len(src))); rc != C.TAD_OK {
		return fmt.Errorf("tad_copy_to_device: %s (code %d)", C.GoString(C.tad_last_error(e.h)), int(rc))
	}
	return nil
}

func (e *Engine) CopyToHost(dst []byte, src unsafe.Pointer) error {
	if len(dst) == 0 {
		return nil
	}
	if rc := C.tad_copy_to_host(e.h, unsafe.Pointer(&dst[0]), src, C.uint64_t(len(dst))); rc != C.TAD_OK {
		return fmt.Errorf("tad_copy_to_host: %s (code %d)", C.GoString(C.tad_last_error(e.h)), int(rc))
	}
	return nil
}

// State is the per-key running EWMA state of a long-running detector (tad.h: tad_state, SURVEY.md 8f rank 3): Spark's streaming moments
// (n, avg, m2), the last EWMA value and the last flowEndSeconds of every key, kept in HBM between batches.
type State struct {
	e *Engine
	h *C.tad_state
}

func (e *Engine) NewState(numKeys uint64) (*State, error) {
	var h *C.tad_state
	if rc := C.tad_state_create(e.h, C.uint64_t(numKeys), &h); rc != C.TAD_OK {
		return nil, fmt.Errorf("tad_state_create: %s (code %d)", C.GoString(C.tad_last_error(e.h)), int(rc))
	}
	return &State{e: e, h: h}, nil
}

func (s *State) Close() {
	if s.h != nil {
		C.tad_state_destroy(s.e.h, s.h)
		s.h = nil
	}
}

// RunStream aggregates ONE new batch and continues every key's recurrences over its new points (tad_run_stream): the rows are the points
// with |x - ewma| > the running stddev_samp.  cols.NumKeys must equal the state's key count; job.Algo must be EWMA.
func (s *State) RunStream(job Job, cols Columns) ([]Row, error) {
	n := len(cols.KeyID)
	if len(cols.FlowEndS) != n || len(cols.Value) != n {
		return nil, errors.New("tadengine: columns differ in length")
	}
	var cj C.tad_job
	cj.algo = C.tad_algo(job.Algo)
	cj.agg_flow = C.tad_agg_flow(job.AggFlow)
	cj.value_op = C.TAD_OP_AUTO
	var cc C.tad_columns
	cc.n_rows = C.uint64_t(n)
	cc.num_keys = C.uint64_t(cols.NumKeys)
	cc.memory = C.TAD_MEM_HOST
	bufs := []unsafe.Pointer{cColumn(cols.KeyID), cColumn(cols.FlowEndS), cColumn(cols.Value)}
	defer func() {
		for _, p := range bufs {
			if p != nil {
				C.free(p)
			}
		}
	}()
	cc.key_id = (*C.uint64_t)(bufs[0])
	cc.flow_end_s = (*C.int64_t)(bufs[1])
	cc.value = (*C.uint64_t)(bufs[2])
	var res *C.tad_result
	if rc := C.tad_run_stream(s.e.h, s.h, &cj, &cc, C.TAD_MEM_HOST, &res); rc != C.TAD_OK {
		msg := C.GoString(C.tad_last_error(s.e.h))
		if rc == C.TAD_ERR_INVALID_ARGUMENT {
			return nil, IllegalArgument{msg}
		}
		return nil, fmt.Errorf("tad_run_stream: %s (code %d)", msg, int(rc))
	}
	defer C.tad_result_free(s.e.h, res)
	a := int(res.n_rows)
	rows := make([]Row, a)
	if a > 0 {
		k := unsafe.Slice((*uint64)(unsafe.Pointer(res.key_id)), a)
		t := unsafe.Slice((*int64)(unsafe.Pointer(res.flow_end_s)), a)
		x := unsafe.Slice((*float64)(unsafe.Pointer(res.throughput)), a)
		c := unsafe.Slice((*float64)(unsafe.Pointer(res.algo_calc)), a)
		sd := unsafe.Slice((*float64)(unsafe.Pointer(res.stddev)), a)
		for i := range rows {
			rows[i] = Row{k[i], t[i], x[i], c[i], sd[i]}
		}
	}
	return rows, nil
}

// Export copies the state to the host: per key the point count, Spark's avg and m2, the last EWMA value and the last flowEndSeconds.
func (s *State) Export(numKeys uint64) (n []uint32, avg, m2, ewma []float64, lastT []int64, err error) {
	n, avg, m2, ewma, lastT = make([]uint32, numKeys), make([]float64, numKeys), make([]float64, numKeys), make([]float64, numKeys), make([]int64, numKeys)
	if numKeys == 0 {
		return
	}
	if rc := C.tad_state_export(s.e.h, s.h, (*C.uint32_t)(unsafe.Pointer(&n[0])), (*C.double)(unsafe.Pointer(&avg[0])), (*C.double)(unsafe.Pointer(&m2[0])),
		(*C.double)(unsafe.Pointer(&ewma[0])), (*C.int64_t)(unsafe.Pointer(&lastT[0]))); rc != C.TAD_OK {
		err = fmt.Errorf("tad_state_export: %s (code %d)", C.GoString(C.tad_last_error(s.e.h)), int(rc))
	}
	return
}

// Progress feeds Status.CompletedStages / TotalStages (controller.go:426-453): the sum over the jobs in flight.
func (e *Engine) Progress() (done, total int) {
	var d, t C.int32_t
	C.tad_progress(e.h, &d, &t)
	return int(d), int(t)
}

// JobProgress is Progress for ONE job: the one whose Job.ID is id (Status.SparkApplication, controller.go:622); total == 0 when no
// such job is in flight.
func (e *Engine) JobProgress(id string) (done, total int) {
	cid := C.CString(id)
	defer C.free(unsafe.Pointer(cid))
	var d, t C.int32_t
	C.tad_job_progress(e.h, cid, &d, &t)
	return int(d), int(t)
}

// JobsInFlight: job contexts busy right now.
func (e *Engine) JobsInFlight() int { return int(C.tad_jobs_in_flight(e.h)) }
