// fluid.js — JavaScript host of the MI355X stable-fluids hot path.
//
// Mirrors the simulation surface of PavelDoGreat/WebGL-Fluid-Simulation's script.js so a caller of
// the reference's globals can switch to this module: same names, same argument meaning, same
// call order into Math.random.  Everything that touched `gl` for simulation in the reference
// (programs, FBOs, blit) is replaced by calls into fluid_napi.node -> libfluid_hip.so (HIP, gfx950).
//
//   reference (script.js)                      here
//   ---------------------------------------   ------------------------------------------------
//   config                       59-85         sim.config            (live plain object, sim keys)
//   canvas.width / canvas.height 56, 1196      sim.canvas            ({width, height} stand-in)
//   pointers, splatStack         100-102       sim.pointers, sim.splatStack
//   initFramebuffers()           982-1010      sim.initFramebuffers()
//   getResolution(resolution)    1612-1624     sim.getResolution(resolution)
//   update()                     1176-1186     sim.update()          (no render; returns dt)
//   calcDeltaTime()              1188-1194     sim.calcDeltaTime()
//   applyInputs()                1219-1229     sim.applyInputs()
//   step(dt)                     1231-1294     sim.step(dt)
//   splatPointer(pointer)        1421-1425     sim.splatPointer(pointer)
//   multipleSplats(amount)       1427-1439     sim.multipleSplats(amount)
//   splat(x, y, dx, dy, color)   1441-1455     sim.splat(x, y, dx, dy, color)
//   correctRadius(radius)        1457-1462     sim.correctRadius(radius)
//   updateColors(dt)             1207-1217     sim.updateColors(dt)
//   mouse / touch / key listeners 1464-1530    sim.dispatch(event)   (headless: events are plain objects)
//   updatePointerDown/Move/UpData 1532-1558    sim.updatePointerDownData / MoveData / UpData
//   correctDeltaX/Y              1560-1570     sim.correctDeltaX / sim.correctDeltaY
//   scaleByPixelRatio(input)     1626-1629     sim.scaleByPixelRatio(input)   (options.pixelRatio, default 1)
//   generateColor()              1565-1571     sim.generateColor()
//   HSVtoRGB(h, s, v)            1573-1595     HSVtoRGB(h, s, v)
//   framebufferToTexture(target) 301-307       sim.framebufferToTexture(target)
//   render(target)               1296-1317     sim.render(target)    (target: {width, height}; bloom, sunrays, shading, blend)
//   captureScreenshot()          287-299       sim.captureScreenshot() -> {width, height, data: Uint8Array RGBA8, top row first}
//   normalizeColor(input)        1597-1604     normalizeColor(input)
//   velocity, dye, pressure, divergence, curl  950-954   sim.velocity ... (width/height/texelSize views)
//
// There is no CPU path: without the addon / a HIP device, createFluid() throws.
//
// Attribution.  The host-side functions in this file that have one natural spelling in JavaScript — the `config` literal, the
// pointer prototype, updatePointerDownData / MoveData / UpData, correctDeltaX / correctDeltaY, splatPointer, multipleSplats,
// generateColor, HSVtoRGB, normalizeColor, wrap, getResolution, scaleByPixelRatio and the listener bodies in dispatch() — follow
// the reference's script.js closely (same names, same statements, `sim.` in front of what were globals) so that the shim is a
// drop-in; they are
//     Copyright (c) 2017 Pavel Dobryakov, MIT License
// and the licence text is reproduced in NOTICE at the repository root.  Everything that talks to the device is original.
'use strict';

const FIELD = { velocity: 0, pressure: 1, divergence: 2, curl: 3, dye: 4 };
const SCHEDULE = { passes: 0, fused: 1 };

function loadBackend () {
    // the native addon; a missing build is an error, not a reason to fall back
    return require('./fluid_napi.node');
}

function defaultConfig () {
    return {
        SIM_RESOLUTION: 128,
        DYE_RESOLUTION: 1024,
        DENSITY_DISSIPATION: 1,
        VELOCITY_DISSIPATION: 0.2,
        PRESSURE: 0.8,
        PRESSURE_ITERATIONS: 20,
        CURL: 30,
        SPLAT_RADIUS: 0.25,
        SPLAT_FORCE: 6000,
        COLORFUL: true,
        COLOR_UPDATE_SPEED: 10,
        PAUSED: false,
        // display compositor (script.js:61, 71-84)
        CAPTURE_RESOLUTION: 512,
        SHADING: true,
        BACK_COLOR: { r: 0, g: 0, b: 0 },
        TRANSPARENT: false,
        BLOOM: true,
        BLOOM_ITERATIONS: 8,
        BLOOM_RESOLUTION: 256,
        BLOOM_INTENSITY: 0.8,
        BLOOM_THRESHOLD: 0.6,
        BLOOM_SOFT_KNEE: 0.7,
        SUNRAYS: true,
        SUNRAYS_RESOLUTION: 196,
        SUNRAYS_WEIGHT: 1.0,
    };
}

function pointerPrototype () {
    this.id = -1;
    this.texcoordX = 0;
    this.texcoordY = 0;
    this.prevTexcoordX = 0;
    this.prevTexcoordY = 0;
    this.deltaX = 0;
    this.deltaY = 0;
    this.down = false;
    this.moved = false;
    this.color = { r: 30, g: 0, b: 300 };
}

function HSVtoRGB (h, s, v) {
    const i = Math.floor(h * 6);
    const f = h * 6 - i;
    const p = v * (1 - s);
    const q = v * (1 - f * s);
    const t = v * (1 - (1 - f) * s);
    const table = [[v, t, p], [q, v, p], [p, v, t], [p, q, v], [t, p, v], [v, p, q]];
    const c = table[i % 6];
    return { r: c[0], g: c[1], b: c[2] };
}

// seedable stand-in for Math.random (the stream BASELINE.md's measurement plan names)
function mulberry32 (seed) {
    let s = seed >>> 0;
    return function () {
        s |= 0; s = s + 0x6D2B79F5 | 0;
        let t = Math.imul(s ^ s >>> 15, 1 | s);
        t = t + Math.imul(t ^ t >>> 7, 61 | t) ^ t;
        return ((t ^ t >>> 14) >>> 0) / 4294967296;
    };
}

function normalizeColor (input) {
    return { r: input.r / 255, g: input.g / 255, b: input.b / 255 };
}

function wrap (value, min, max) {
    const range = max - min;
    if (range == 0) return min;
    return (value - min) % range + min;
}

// options: { canvas: {width, height}, config: {...overrides}, device, schedule: 'fused'|'passes', storage: 'f32'|'f16',
//            random: () => number (defaults to Math.random), seed: number (shorthand for random: mulberry32(seed)), backend: <object with the addon's functions>,
//            tile: { rank, world, tilesX = 1, halo = 56, commId: Buffer } }
// `storage`: 'f32' keeps fp32 fields (what the headless reference build keeps); 'f16' keeps half texels like the reference's
// half-float textures on a real GPU (ext.halfFloatTexType): every pass output is rounded to fp16, a step moves half the bytes.
// `tile`: this process is one rank of a multi-GPU run (one process per GPU).  The rank owns row stripe
// floor(rank / tilesX) of world / tilesX, column tile rank % tilesX; rank 0 creates the id with commUniqueId() and
// ships it to the other ranks (file, env, socket — any transport); after that step() exchanges ghost rows / columns with
// its neighbours by ncclSend / ncclRecv inside libfluid_hip.so.  Every rank issues the same splats (same seed).
function createFluid (options) {
    options = options || {};
    const native = options.backend || loadBackend();
    // `seed` (a number) is the serialisable way to ask for a reproducible Math.random: options.random wins if both are given
    const random = options.random || (options.seed !== undefined ? mulberry32(options.seed) : Math.random);
    const sim = {};

    sim.config = Object.assign(defaultConfig(), options.config || {});
    sim.canvas = Object.assign({ width: 512, height: 512 }, options.canvas || {});
    sim.pointers = [new pointerPrototype()];
    sim.splatStack = [];

    let handle = null;
    let lastUpdateTime = Date.now();
    let colorUpdateTimer = 0.0;
    const schedule = SCHEDULE[options.schedule || 'fused'];
    const device = options.device || 0;
    const storage = { f32: 0, f16: 1 }[options.storage || 'f32'];
    if (storage === undefined) throw new Error("fluid: storage must be 'f32' or 'f16'");

    function fieldView (name, isDouble) {
        const view = {
            get width () { return native.fieldInfo(handle, FIELD[name]).width; },
            get height () { return native.fieldInfo(handle, FIELD[name]).height; },
            get texelSizeX () { return 1.0 / this.width; },
            get texelSizeY () { return 1.0 / this.height; },
            name,
        };
        if (isDouble) {
            // the library swaps read/write internally; `.read` always names the current read side
            Object.defineProperty(view, 'read', { get () { return view; } });
            view.swap = function () {};
        }
        return view;
    }
    sim.velocity = fieldView('velocity', true);
    sim.dye = fieldView('dye', true);
    sim.pressure = fieldView('pressure', true);
    sim.divergence = fieldView('divergence', false);
    sim.curl = fieldView('curl', false);

    sim.getResolution = function (resolution) {
        let aspectRatio = sim.canvas.width / sim.canvas.height;
        if (aspectRatio < 1) aspectRatio = 1.0 / aspectRatio;
        const min = Math.round(resolution);
        const max = Math.round(resolution * aspectRatio);
        if (sim.canvas.width > sim.canvas.height) return { width: max, height: min };
        return { width: min, height: max };
    };

    // dye and velocity survive a resolution change (bilinear copy); pressure / divergence / curl restart at 0
    sim.initFramebuffers = function () {
        const simRes = sim.getResolution(sim.config.SIM_RESOLUTION);
        const dyeRes = sim.getResolution(sim.config.DYE_RESOLUTION);
        if (handle == null && options.tile) {
            const t = options.tile, tilesX = t.tilesX || 1;
            handle = native.createTile(simRes.width, simRes.height, dyeRes.width, dyeRes.height, device, schedule,
                Math.floor(t.rank / tilesX), t.world / tilesX, t.rank % tilesX, tilesX, t.halo === undefined ? 56 : t.halo, storage);
            if (t.reach !== undefined) native.setReach(handle, t.reach);
            if (Array.isArray(t.linkModel)) native.setLinkModel(handle, t.linkModel[0], t.linkModel[1]);   // [latency us, GB/s] of one neighbour message
            // FLUID_TRACE_COMM: marker lines on stderr around the one call that talks to RCCL (ncclCommInitRank), so that a wedged
            // communicator bootstrap on a box can be told from a hang in this library (tests/test_node_shim.py)
            const trace = !!process.env.FLUID_TRACE_COMM;
            if (trace) process.stderr.write('[fluid.js] commInit begin rank ' + t.rank + '/' + t.world + '\n');
            native.commInit(handle, t.commId);          // collective: every rank of the run calls it
            if (trace) process.stderr.write('[fluid.js] commInit done rank ' + t.rank + '\n');
            // linkModel: [latency us, GB/s] given, or 'calibrate' — measured now, on this set's own links (collective like commInit;
            // what came out is sim.linkModel).  A probe that fails leaves the library's constants in place.
            if (t.linkModel === 'calibrate' && t.world > 1) {
                try { sim.linkModel = native.calibrateLink(handle, 20); } catch (e) { sim.linkModel = null; sim.linkModelError = String(e.message || e); }
            }
        } else if (handle == null) handle = native.create(simRes.width, simRes.height, dyeRes.width, dyeRes.height, device, schedule, storage);
        else native.resize(handle, simRes.width, simRes.height, dyeRes.width, dyeRes.height);
    };

    sim.correctRadius = function (radius) {
        const aspectRatio = sim.canvas.width / sim.canvas.height;
        if (aspectRatio > 1) radius *= aspectRatio;
        return radius;
    };

    sim.splat = function (x, y, dx, dy, color) {
        native.splat(handle, x, y, dx, dy, color.r, color.g, color.b,
            sim.canvas.width / sim.canvas.height, sim.correctRadius(sim.config.SPLAT_RADIUS / 100.0));
    };

    sim.generateColor = function () {
        const c = HSVtoRGB(random(), 1.0, 1.0);
        c.r *= 0.15;
        c.g *= 0.15;
        c.b *= 0.15;
        return c;
    };

    sim.multipleSplats = function (amount) {
        for (let i = 0; i < amount; i++) {
            const color = sim.generateColor();
            color.r *= 10.0;
            color.g *= 10.0;
            color.b *= 10.0;
            const x = random();
            const y = random();
            const dx = 1000 * (random() - 0.5);
            const dy = 1000 * (random() - 0.5);
            sim.splat(x, y, dx, dy, color);
        }
    };

    sim.splatPointer = function (pointer) {
        const dx = pointer.deltaX * sim.config.SPLAT_FORCE;
        const dy = pointer.deltaY * sim.config.SPLAT_FORCE;
        sim.splat(pointer.texcoordX, pointer.texcoordY, dx, dy, pointer.color);
    };

    sim.applyInputs = function () {
        if (sim.splatStack.length > 0) sim.multipleSplats(sim.splatStack.pop());
        sim.pointers.forEach(p => {
            if (p.moved) {
                p.moved = false;
                sim.splatPointer(p);
            }
        });
    };

    sim.updateColors = function (dt) {
        if (!sim.config.COLORFUL) return;
        colorUpdateTimer += dt * sim.config.COLOR_UPDATE_SPEED;
        if (colorUpdateTimer >= 1) {
            colorUpdateTimer = wrap(colorUpdateTimer, 0, 1);
            sim.pointers.forEach(p => { p.color = sim.generateColor(); });
        }
    };

    // ---- input path (SURVEY §8f N2): the reference's DOM listeners, headless --------------------------------
    const pixelRatio = options.pixelRatio || 1;
    sim.scaleByPixelRatio = function (input) { return Math.floor(input * pixelRatio); };

    sim.correctDeltaX = function (delta) {
        const aspectRatio = sim.canvas.width / sim.canvas.height;
        if (aspectRatio < 1) delta *= aspectRatio;
        return delta;
    };

    sim.correctDeltaY = function (delta) {
        const aspectRatio = sim.canvas.width / sim.canvas.height;
        if (aspectRatio > 1) delta /= aspectRatio;
        return delta;
    };

    sim.updatePointerDownData = function (pointer, id, posX, posY) {
        pointer.id = id;
        pointer.down = true;
        pointer.moved = false;
        pointer.texcoordX = posX / sim.canvas.width;
        pointer.texcoordY = 1.0 - posY / sim.canvas.height;
        pointer.prevTexcoordX = pointer.texcoordX;
        pointer.prevTexcoordY = pointer.texcoordY;
        pointer.deltaX = 0;
        pointer.deltaY = 0;
        pointer.color = sim.generateColor();
    };

    sim.updatePointerMoveData = function (pointer, posX, posY) {
        pointer.prevTexcoordX = pointer.texcoordX;
        pointer.prevTexcoordY = pointer.texcoordY;
        pointer.texcoordX = posX / sim.canvas.width;
        pointer.texcoordY = 1.0 - posY / sim.canvas.height;
        pointer.deltaX = sim.correctDeltaX(pointer.texcoordX - pointer.prevTexcoordX);
        pointer.deltaY = sim.correctDeltaY(pointer.texcoordY - pointer.prevTexcoordY);
        pointer.moved = Math.abs(pointer.deltaX) > 0 || Math.abs(pointer.deltaY) > 0;
    };

    sim.updatePointerUpData = function (pointer) {
        pointer.down = false;
    };

    // One recorded DOM event: { type, offsetX, offsetY } (mouse), { type, touches: [{identifier, pageX, pageY}] }
    // (touchstart / touchmove: targetTouches; touchend: changedTouches), { type: 'keydown', code, key }.
    // Same bodies as the listeners of script.js:1464-1530.
    sim.dispatch = function (e) {
        const pointers = sim.pointers;
        switch (e.type) {
        case 'mousedown': {
            const posX = sim.scaleByPixelRatio(e.offsetX);
            const posY = sim.scaleByPixelRatio(e.offsetY);
            let pointer = pointers.find(p => p.id == -1);
            if (pointer == null) pointer = new pointerPrototype();
            sim.updatePointerDownData(pointer, -1, posX, posY);
            break;
        }
        case 'mousemove': {
            const pointer = pointers[0];
            if (!pointer.down) return;
            sim.updatePointerMoveData(pointer, sim.scaleByPixelRatio(e.offsetX), sim.scaleByPixelRatio(e.offsetY));
            break;
        }
        case 'mouseup':
            sim.updatePointerUpData(pointers[0]);
            break;
        case 'touchstart': {
            const touches = e.touches;
            while (touches.length >= pointers.length) pointers.push(new pointerPrototype());
            for (let i = 0; i < touches.length; i++) {
                sim.updatePointerDownData(pointers[i + 1], touches[i].identifier,
                    sim.scaleByPixelRatio(touches[i].pageX), sim.scaleByPixelRatio(touches[i].pageY));
            }
            break;
        }
        case 'touchmove': {
            const touches = e.touches;
            for (let i = 0; i < touches.length; i++) {
                const pointer = pointers[i + 1];
                if (!pointer.down) continue;
                sim.updatePointerMoveData(pointer, sim.scaleByPixelRatio(touches[i].pageX), sim.scaleByPixelRatio(touches[i].pageY));
            }
            break;
        }
        case 'touchend': {
            const touches = e.touches;
            for (let i = 0; i < touches.length; i++) {
                const pointer = pointers.find(p => p.id == touches[i].identifier);
                if (pointer == null) continue;
                sim.updatePointerUpData(pointer);
            }
            break;
        }
        case 'keydown':
            if (e.code === 'KeyP') sim.config.PAUSED = !sim.config.PAUSED;
            if (e.key === ' ') sim.splatStack.push(parseInt(random() * 20) + 5);
            break;
        default:
            throw new Error('fluid.js: unknown event type ' + e.type);
        }
    };

    // replay a recorded session: frames = [{ dt, events: [...] }]; per frame the events are dispatched and one
    // update() runs with that frame's dt (clamped like calcDeltaTime)
    sim.replay = function (frames) {
        frames.forEach(f => {
            (f.events || []).forEach(sim.dispatch);
            sim.update(f.dt);
        });
    };

    sim.calcDeltaTime = function () {
        const now = Date.now();
        let dt = (now - lastUpdateTime) / 1000;
        dt = Math.min(dt, 0.016666);
        lastUpdateTime = now;
        return dt;
    };

    // one reference step(dt) with the CURRENT config values (they are read every step, like the reference)
    sim.step = function (dt, n) {
        const c = sim.config;
        native.step(handle, n === undefined ? 1 : n, dt, c.CURL, c.PRESSURE, c.PRESSURE_ITERATIONS,
            c.VELOCITY_DISSIPATION, c.DENSITY_DISSIPATION);
    };

    // update() without render(null) / requestAnimationFrame: the caller owns the frame loop
    sim.update = function (fixedDt) {
        const dt = fixedDt === undefined ? sim.calcDeltaTime() : Math.min(fixedDt, 0.016666);
        sim.updateColors(dt);
        sim.applyInputs();
        if (!sim.config.PAUSED) sim.step(dt);
        return dt;
    };

    // readPixels(RGBA, FLOAT): R and RG targets come back padded to (r, g, 0, 1); row 0 = bottom
    sim.framebufferToTexture = function (target) {
        if (target && target.isFrame) return native.readFrame(handle, target.width, target.height);
        const name = typeof target === 'string' ? target : target.name;
        const info = native.fieldInfo(handle, FIELD[name]);
        const src = native.readField(handle, FIELD[name]);
        if (info.channels === 4) return src;
        const n = info.width * info.height, out = new Float32Array(n * 4);
        for (let i = 0; i < n; i++) {
            for (let k = 0; k < info.channels; k++) out[4 * i + k] = src[info.channels * i + k];
            out[4 * i + 3] = 1.0;
        }
        return out;
    };

    // ---- display compositor (SURVEY §8f N3) ---------------------------------------------------------------------
    // render(target), script.js:1296-1317: target = { width, height } stands for the float FBO captureScreenshot()
    // creates; the frame stays on the device until framebufferToTexture(target) / captureScreenshot() read it
    sim.render = function (target) {
        const c = sim.config;
        const bloomRes = sim.getResolution(c.BLOOM_RESOLUTION);
        const sunRes = sim.getResolution(c.SUNRAYS_RESOLUTION);
        const back = normalizeColor(c.BACK_COLOR);
        native.render(handle, target.width, target.height, c.SHADING ? 1 : 0, c.BLOOM ? 1 : 0, c.SUNRAYS ? 1 : 0, c.TRANSPARENT ? 1 : 0,
            back.r, back.g, back.b, bloomRes.width, bloomRes.height, c.BLOOM_ITERATIONS, c.BLOOM_INTENSITY, c.BLOOM_THRESHOLD,
            c.BLOOM_SOFT_KNEE, sunRes.width, sunRes.height, c.SUNRAYS_WEIGHT);
        target.isFrame = true;
    };

    // captureScreenshot(), script.js:287-299, up to the PNG encoder: render into a CAPTURE_RESOLUTION target,
    // framebufferToTexture + normalizeTexture (clamp01 * 255, rows flipped: top row first)
    sim.captureScreenshot = function () {
        const res = sim.getResolution(sim.config.CAPTURE_RESOLUTION);
        const target = { width: res.width, height: res.height };
        sim.render(target);
        return { width: res.width, height: res.height, data: native.readFrameRgba8(handle, res.width, res.height) };
    };

    // the dithering texture's R channel in [0, 1] (script.js:958; default = the 1 x 1 white placeholder of 1135)
    sim.setDitheringTexture = function (r, width, height) { native.setDither(handle, r, width, height); };

    sim.readField = function (name) { return native.readField(handle, FIELD[name]); };
    sim.writeField = function (name, data) { native.writeField(handle, FIELD[name], data); };
    sim.sync = function () { native.sync(handle); };
    sim.checkHalo = function () { native.haloCheck(handle); };         // multi-GPU: throws if a back-trace outran the ghost rows
    sim.exchangeCount = function () { return native.exchangeCount(handle); };
    sim.setTiming = function (on) { native.setTiming(handle, on ? 1 : 0); };
    // the page never looks at the curl texture outside step() (script.js:1234-1243): a host that does not either saves its store per call
    sim.setCurlOutput = function (on) { native.setCurlOutput(handle, on ? 1 : 0); };
    sim.getTimings = function () { return native.getTimings(handle); };
    // what the next step(dt, n) would launch with the CURRENT config (nothing runs): { fused, jacobiLaunches, chained, runsAhead, dyePacked, ... }
    sim.scheduleInfo = function (dt, n) {
        const c = sim.config;
        return native.scheduleInfo(handle, n === undefined ? 1 : n, dt, c.CURL, c.PRESSURE, c.PRESSURE_ITERATIONS,
            c.VELOCITY_DISSIPATION, c.DENSITY_DISSIPATION);
    };
    // device time of each step of the next step(dt, n) calls, without a sync per step (the first `capacity` steps of a call; 0 = off)
    sim.setStepMarks = function (capacity) { native.setStepMarks(handle, capacity); };
    sim.stepMarks = function () { return native.getStepMarks(handle); };
    sim.destroy = function () { if (handle != null) { native.destroy(handle); handle = null; } };

    sim.initFramebuffers();
    return sim;
}

function commUniqueId () { return loadBackend().commUniqueId(); }

module.exports = { createFluid, commUniqueId, HSVtoRGB, normalizeColor, mulberry32, pointerPrototype, defaultConfig, FIELD, SCHEDULE };
