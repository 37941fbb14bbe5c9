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
// (controller.go:199-201); calls on one engine serialise inside the library.
type Engine struct{ h *C.tad_engine }

func NewEngine(device int) (*Engine, error) {
	opts := C.tad_engine_opts{device: C.int32_t(device)}
	var h *C.tad_engine
	if rc := C.tad_engine_create(&opts, &h); rc != C.TAD_OK {
		return nil, fmt.Errorf("tad_engine_create: %s (code %d)", C.GoString(C.tad_last_error(nil)), int(rc))
	}
	e := &Engine{h: h}
	runtime.SetFinalizer(e, func(e *Engine) { e.Close() })
	return e, nil
}

// SetPlan replaces the engine's plan overrides (tad.h: tad_plan, ABI 7; the zero value = the engine decides, which is what
// the controller uses).  Tests and A/B measurements force a strategy with it; the library reads no environment variable.
func (e *Engine) SetPlan(p C.tad_plan) error {
	if rc := C.tad_engine_set_plan(e.h, &p); rc != C.TAD_OK {
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
		int32(res.stats.stage0_path)}
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

// Progress feeds Status.CompletedStages / TotalStages (controller.go:426-453).
func (e *Engine) Progress() (done, total int) {
	var d, t C.int32_t
	C.tad_progress(e.h, &d, &t)
	return int(d), int(t)
}
