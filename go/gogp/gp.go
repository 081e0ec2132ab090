// Package gogp is the drop-in replacement of bitbucket.org/dtolpin/gogp/gp for the
// hot path (Absorb / LML / Produce / Observe / Gradient), calling the MI355X HIP
// library through cgo.  NOT COMPILED IN THIS REPOSITORY'S CI: the build image has
// no Go toolchain; the same C ABI is exercised from Python (ctypes) and C++.
//
// Build:  CGO_CFLAGS="-I${REPO}/include" CGO_LDFLAGS="-L${REPO}/gogp_amd -lgogp_hip" go build ./go/gogp
package gogp

/*
#include <stdlib.h>
#include "gogp_hip.h"
*/
import "C"

import (
	"fmt"
	"math"
	"runtime"
	"unsafe"
)

// Kernel is the reference's kernel interface (gp/gp.go:14-17) plus the closed
// description the device path needs.  Kernels that cannot describe themselves
// (arbitrary Go code) must keep using the reference package.
type Kernel interface {
	Observe([]float64) float64
	NTheta() int
}

// DeviceKernel is implemented by similarity kernels that can run on the GPU.
type DeviceKernel interface {
	Kernel
	Terms() []Term
}

// DeviceNoise is implemented by noise kernels that can run on the GPU.
type DeviceNoise interface {
	Kernel
	NoiseDesc() (kind int, std, scale float64)
}

// Term mirrors gogp_term of include/gogp_hip.h.
type Term struct {
	Kind, ScaleIdx, LenIdx int
	ARD                    bool
	PeriodIdx              int
	PeriodMult             float64
}

// GP keeps the reference's exported fields (gp/gp.go:20-38).
type GP struct {
	NDim         int
	Simil, Noise Kernel

	ThetaSimil, ThetaNoise []float64
	X                      [][]float64
	Y                      []float64

	Parallel bool // accepted for compatibility; the device path is always parallel

	// Cached computations (gp/gp.go:34-36).  Alpha is refreshed by every Absorb / Observe;
	// the factor is fetched on demand by Factor() (n*n doubles over PCIe are not free).
	Alpha []float64 // K^-1 y

	h       *C.gogp_handle
	withObs bool
	lastLen int
	// X, Y are uploaded only when they changed.  Absorb and the full form of Observe assign
	// them; code that writes gp.X / gp.Y directly (the reference's tutorial does:
	// tutorial/tutorial.go:114-115) is detected by comparing the slice headers below, and
	// in-place edits of the same backing arrays must call Touch().
	dirty      bool
	upX        *[]float64 // &gp.X[0] at the time of the last upload
	upY        *float64   // &gp.Y[0] at the time of the last upload
	upN        int
}

// Touch marks X / Y as modified in place: the next Absorb / Observe uploads them again.
func (gp *GP) Touch() { gp.dirty = true }

func (gp *GP) handle() *C.gogp_handle {
	if gp.h != nil {
		return gp.h
	}
	sim, ok := gp.Simil.(DeviceKernel)
	if !ok {
		panic("gogp: Simil does not implement DeviceKernel; use the reference package for arbitrary Go kernels")
	}
	var d C.gogp_desc
	d.ndim = C.int32_t(gp.NDim)
	terms := sim.Terms()
	d.nterms = C.int32_t(len(terms))
	d.ntheta_simil = C.int32_t(sim.NTheta())
	for i, t := range terms {
		ct := &d.terms[i]
		ct.kind, ct.scale_idx, ct.len_idx = C.int32_t(t.Kind), C.int32_t(t.ScaleIdx), C.int32_t(t.LenIdx)
		if t.ARD {
			ct.ard = 1
		}
		ct.period_idx, ct.period_mult = C.int32_t(t.PeriodIdx), C.double(t.PeriodMult)
	}
	if gp.Noise == nil { // gp/gp.go:45-48: ConstantNoise(1e-5)
		d.noise_kind, d.noise_std = C.GOGP_NOISE_CONSTANT, 1e-5
	} else {
		dn, ok := gp.Noise.(DeviceNoise)
		if !ok {
			panic("gogp: Noise does not implement DeviceNoise; use the reference package for arbitrary Go kernels")
		}
		kind, std, scale := dn.NoiseDesc()
		d.noise_kind, d.noise_std, d.noise_scale = C.int32_t(kind), C.double(std), C.double(scale)
	}
	if rc := C.gogp_create(&d, -1, &gp.h); rc != C.GOGP_OK {
		panic(fmt.Sprintf("gogp_create: %s", C.GoString(C.gogp_last_error(nil))))
	}
	runtime.SetFinalizer(gp, func(g *GP) { C.gogp_destroy(g.h) })
	gp.dirty = true
	return gp.h
}

func (gp *GP) err(rc C.int) error {
	if rc == C.GOGP_OK {
		return nil
	}
	return fmt.Errorf("gogp_hip: %s", C.GoString(C.gogp_last_error(gp.h)))
}

func dptr(s []float64) *C.double {
	if len(s) == 0 {
		return nil
	}
	return (*C.double)(unsafe.Pointer(&s[0]))
}

// changed reports whether gp.X / gp.Y were re-assigned since the last upload.
func (gp *GP) changed() bool {
	if gp.dirty || len(gp.X) != gp.upN || len(gp.Y) != gp.upN {
		return true
	}
	if gp.upN == 0 {
		return false
	}
	return &gp.X[0] != gp.upX || &gp.Y[0] != gp.upY
}

// pushData packs X ([][]float64, not contiguous) row-major and copies X, Y to the device --
// only when they changed: a hyperparameter step must not pay an O(N*D) host-to-device copy
// and a drain of the device streams.
func (gp *GP) pushData() error {
	h := gp.handle()
	if !gp.changed() {
		return nil
	}
	n := len(gp.X)
	if len(gp.Y) != n {
		return fmt.Errorf("gogp: len(X) != len(Y)")
	}
	flat := make([]float64, n*gp.NDim)
	for i, row := range gp.X {
		copy(flat[i*gp.NDim:], row)
	}
	if err := gp.err(C.gogp_set_data(h, dptr(flat), dptr(gp.Y), C.int64_t(n))); err != nil {
		return err
	}
	gp.dirty, gp.upN = false, n
	if n > 0 {
		gp.upX, gp.upY = &gp.X[0], &gp.Y[0]
	}
	return nil
}

func (gp *GP) defaults() { // gp/gp.go:45-57
	if len(gp.ThetaSimil) == 0 {
		gp.ThetaSimil = make([]float64, gp.Simil.NTheta())
	}
	nn := 0
	if gp.Noise != nil {
		nn = gp.Noise.NTheta()
	}
	if len(gp.ThetaNoise) == 0 {
		gp.ThetaNoise = make([]float64, nn)
	}
}

// Absorb absorbs observations into the process (gp/gp.go:80-87).
func (gp *GP) Absorb(x [][]float64, y []float64) (err error) {
	gp.defaults()
	gp.X, gp.Y = x, y
	if err = gp.pushData(); err != nil {
		return err
	}
	tn := gp.ThetaNoise
	if len(tn) == 0 {
		tn = []float64{0}
	}
	rc := C.gogp_absorb(gp.handle(), dptr(gp.ThetaSimil), dptr(tn))
	if rc != C.GOGP_OK && rc != C.GOGP_ECOND {
		return gp.err(rc) // gp/gp.go:228-230
	}
	if err = gp.fetchState(); err != nil {
		return err
	}
	return gp.err(rc) // GOGP_ECOND: gonum's Condition error, returned as gp/gp.go:233-236 does
}

// Factor returns the lower Cholesky factor, row-major n x n (gonum's mat.Cholesky keeps
// U = L^T: gp.GP.L of the reference, gp/gp.go:35).
func (gp *GP) Factor() ([]float64, error) {
	n := int(C.gogp_n(gp.handle()))
	L := make([]float64, n*n)
	if err := gp.err(C.gogp_get_factor(gp.h, dptr(L))); err != nil {
		return nil, err
	}
	return L, nil
}

func (gp *GP) fetchState() error {
	n := int(C.gogp_n(gp.h))
	gp.Alpha = make([]float64, n)
	return gp.err(C.gogp_get_alpha(gp.h, dptr(gp.Alpha)))
}

// LML computes log marginal likelihood (gp/gp.go:244-253).
func (gp *GP) LML() float64 {
	var v C.double
	if err := gp.err(C.gogp_lml(gp.handle(), &v)); err != nil {
		panic(err)
	}
	return float64(v)
}

// Produce computes predictions (gp/gp.go:258-360).
func (gp *GP) Produce(x [][]float64) (mu, sigma []float64, err error) {
	gp.defaults()
	m := len(x)
	flat := make([]float64, m*gp.NDim)
	for i, row := range x {
		copy(flat[i*gp.NDim:], row)
	}
	mu, sigma = make([]float64, m), make([]float64, m)
	if err = gp.err(C.gogp_produce(gp.handle(), dptr(flat), C.int64_t(m), dptr(mu), dptr(sigma))); err != nil {
		return nil, nil, err // gp/gp.go:338-340
	}
	return mu, sigma, nil
}

// Observe computes the log marginal likelihood of log-transformed
// hyperparameters [, inputs, outputs] (gp/gp.go:374-413).  Panics where the
// reference panics.
func (gp *GP) Observe(x []float64) float64 {
	gp.defaults()
	P := len(gp.ThetaSimil) + len(gp.ThetaNoise)
	var lml C.double
	var rc C.int
	if len(x) == P {
		if err := gp.pushData(); err != nil {
			panic(err)
		}
		gp.withObs = false
		rc = C.gogp_observe(gp.handle(), dptr(x), C.int64_t(len(x)), &lml)
	} else {
		rest := len(x) - P
		if rest < 0 || rest%(gp.NDim+1) != 0 {
			panic("len(x)") // gp/gp.go:398-400
		}
		gp.withObs = true
		rc = C.gogp_observe_full(gp.handle(), dptr(x), C.int64_t(len(x)), &lml)
		n := rest / (gp.NDim + 1) // gp/gp.go:391-396: X, Y re-sliced from x
		gp.X = make([][]float64, n)
		for i := range gp.X {
			gp.X[i] = x[P+i*gp.NDim : P+(i+1)*gp.NDim]
		}
		gp.Y = x[P+n*gp.NDim:]
		// the device holds exactly these data now (gogp_observe_full replaced them), unless the
		// call failed half-way
		gp.dirty, gp.upN = rc != C.GOGP_OK, n
		if n > 0 {
			gp.upX, gp.upY = &gp.X[0], &gp.Y[0]
		}
	}
	if err := gp.err(rc); err != nil {
		panic(err) // gp/gp.go:402-405 (also gonum's Condition error: GOGP_ECOND)
	}
	if err := gp.fetchState(); err != nil { // Alpha is part of the documented state (gp/gp.go:255-257)
		panic(err)
	}
	for i := range gp.ThetaSimil { // gp/gp.go:384-385
		gp.ThetaSimil[i] = math.Exp(x[i])
	}
	for i := range gp.ThetaNoise {
		gp.ThetaNoise[i] = math.Exp(x[len(gp.ThetaSimil)+i])
	}
	gp.lastLen = len(x)
	return float64(lml)
}

// ObserveGradientCandidates evaluates k candidate log-theta vectors (hyperparameters-only form)
// on the GP's data in ONE launch sequence and returns what Observe + Gradient would return for
// each; the GP's own state is not touched.  status[c] is C.GOGP_ENOTPD where the matrix of
// candidate c is not positive definite (lml NaN, gradient zeros).  This is the device-side
// form of optimize.Settings.Concurrent (tutorial/tutorial.go:30,141): one GP instead of NTASKS.
func (gp *GP) ObserveGradientCandidates(xs [][]float64) (lml []float64, grad [][]float64, status []int) {
	k, p := len(xs), len(gp.ThetaSimil)+len(gp.ThetaNoise)
	flat := make([]float64, k*p)
	for c, x := range xs {
		if len(x) != p {
			panic("len(x)") // gp/gp.go:398-400
		}
		copy(flat[c*p:], x)
	}
	if err := gp.pushData(); err != nil {
		panic(err)
	}
	lml = make([]float64, k)
	gflat := make([]float64, k*p)
	st := make([]C.int, k)
	rc := C.gogp_observe_gradient_candidates(gp.handle(), C.int(k), dptr(flat), C.int64_t(p),
		dptr(lml), dptr(gflat), &st[0])
	if rc != C.GOGP_OK && rc != C.GOGP_ENOTPD && rc != C.GOGP_ECOND {
		panic(gp.err(rc))
	}
	grad, status = make([][]float64, k), make([]int, k)
	for c := range grad {
		grad[c] = gflat[c*p : (c+1)*p]
		status[c] = int(st[c])
	}
	return lml, grad, status
}

// SetOption sets a schedule / precision option of the device path (gogp_set_option; no reference
// counterpart).  "gradient_precision" = 32 keeps Observe's LML, Alpha and Produce in fp64 and runs
// what only Gradient needs (the triangular inverse and K^-1) on the fp32 matrix cores: the gradient
// stays within ~1e-8 of the fp64 one (the reference checks its own to 1e-4, gp/gp_test.go:170,248)
// at 19 instead of 14 evaluations per second at N = 16384.  Only for kernels of ONE term with an output
// scale: for a sum of terms the library refuses the option (an error) rather than return scale components
// read off the float K^-1 (1.9e-4 measured).
func (gp *GP) SetOption(name string, value int64) error {
	cn := C.CString(name)
	defer C.free(unsafe.Pointer(cn))
	return gp.err(C.gogp_set_option(gp.handle(), cn, C.int64_t(value)))
}

// Gradient computes the gradient of the log-likelihood (gp/gp.go:418-499).
func (gp *GP) Gradient() []float64 {
	grad := make([]float64, gp.lastLen)
	if err := gp.err(C.gogp_gradient(gp.handle(), dptr(grad), C.int64_t(len(grad)))); err != nil {
		panic(err)
	}
	return grad
}
